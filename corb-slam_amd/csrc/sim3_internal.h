// sim3_internal.h -- device argument block of the fused OptimizeSim3 kernel (sim3_kernels.hip / corb_sim3.cpp).
#pragma once
#include "corb_internal.h"

struct CorbSim3Dev {
    int n_problems;
    const int* off;                   // [n_problems + 1] correspondence range of each problem
    const float* p1c; const float* p2c;            // [N][3]
    const float* obs1; const float* obs2;          // [N][2]
    const float* w1; const float* w2;              // [N]
    const float* K;                   // [n_problems][8] fx1 fy1 cx1 cy1 fx2 fy2 cx2 cy2
    double* S;                        // [n_problems][8] quaternion xyzw, t, s -- in: start (quaternion from the rotation matrix), out: result
    unsigned char* removed;           // [N]
    double* last12; double* last21;   // [N]
    int* counters;                    // [n_problems][4] iterations, trials, nIn, updated
    float th2; int fix_scale;
};

void sim3_launch_optimize(const CorbSim3Dev& d, hipStream_t s);
