// placeholder until ba_kernels.hip lands: fails loudly (never a CPU fallback)
#include "corb_internal.h"
void corb_set_error(const char* fmt, ...);
extern "C" int corb_ba_solve(const CorbBAProblem*, int, int, volatile int*, CorbBAResult*, int)
{
    corb_set_error("corb_ba_solve: not built in this revision");
    return CORB_ERR_ARG;
}
