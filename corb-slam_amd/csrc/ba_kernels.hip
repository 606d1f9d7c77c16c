// ba_kernels.hip -- global bundle adjustment kernels (to be filled in)
#include "corb_internal.h"
