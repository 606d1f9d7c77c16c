#include "orc.h"
int orc_ba_solve(const OrcBAProblem* p, int iters, int robust, volatile int* stop, OrcBAResult* r){(void)p;(void)iters;(void)robust;(void)stop;(void)r;return -1;}
