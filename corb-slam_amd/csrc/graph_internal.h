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
    // deterministic accumulation of the normal equations: the edge kernel stores its Jacobians and error, then one wavefront per free vertex
    // (diagonal block + right-hand side) and one per connected vertex pair (off-diagonal block) sum their edges in edge order
    double* ejac;                 // [E][105] J_i (49, row-major [r][dof]), J_j (49), error (7)
    const int* voff; const int* vedge;            // [nP+1], [.] incident edges of free vertex h: edge id << 1 | role (0 = vertex i of the edge)
    int n_pairs; const int* poff; const int* pedge;   // [n_pairs+1], [.] edges of pair g: edge id << 1 | flipped (edge's (i, j) is the pair's (hi, lo))
    const int* plo; const int* phi;               // [n_pairs] hessian indices lo < hi
};

void eg_launch_chi2(const CorbGraphDev& d, int nparts, double* out, hipStream_t s);
void eg_launch_build(const CorbGraphDev& d, hipStream_t s);
void eg_launch_lambda(const CorbGraphDev& d, double lambda, hipStream_t s);
void eg_launch_update(const CorbGraphDev& d, double lambda, double* scale_out, hipStream_t s);
void eg_launch_apply(int K, const double* S_old, const double* S_new, float* Tiw, int M, const int* ref, float* points, hipStream_t s);
