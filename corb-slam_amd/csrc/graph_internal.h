// graph_internal.h -- device argument block and launchers of the essential-graph optimisation (graph_kernels.hip / corb_graph.cpp).
#pragma once
#include "corb_internal.h"

struct CorbGraphDev {
    int K, E, nP, sp, fix_scale;
    double* V;                    // [K][8]
    const unsigned char* fixed;   // [K]
    const int* idx;               // [K] hessian index or -1
    const int* vi; const int* vj; // [E]
    const double* meas;           // [E][8]
    double* H; double* A; double* b; double* x;
    double* partial;              // block partial sums
};

void eg_launch_chi2(const CorbGraphDev& d, int nparts, double* out, hipStream_t s);
void eg_launch_build(const CorbGraphDev& d, hipStream_t s);
void eg_launch_lambda(const CorbGraphDev& d, double lambda, hipStream_t s);
void eg_launch_update(const CorbGraphDev& d, double lambda, double* scale_out, hipStream_t s);
void eg_launch_apply(int K, const double* S_old, const double* S_new, float* Tiw, int M, const int* ref, float* points, hipStream_t s);
