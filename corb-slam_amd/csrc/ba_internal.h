// ba_internal.h -- device argument block of the bundle-adjustment kernels.
#pragma once
#include "corb_internal.h"

#define BA_EDGE_STRIDE 54     // doubles per edge: A'WA(6) -A'We(3) B'WB(21) -B'We(6) B'WA(18)

struct CorbBADev {
    int nE, nP, nL, sp;           // active edges, free poses, free landmarks, 6*nP
    int robust;
    double fx, fy, cx, cy, bf, delta2, delta3;
    // edges, sorted by landmark (free-pose edges first inside a landmark); edges of fixed landmarks last
    const int* e_pose; const int* e_point;        // hessian indices (-1 = fixed vertex)
    const int* e_vpose; const int* e_vpoint;      // vertex indices
    const double* e_obs; const double* e_w; const unsigned char* e_dim;
    const int* loff;              // [nL+1] edge range of each free landmark
    const int* lnfree;            // [nL]   number of free-pose edges (the leading ones)
    const int* poff; const int* pedge;            // per free pose: edge ids
    const int* pose_vertex; const int* point_vertex;
    double* pose_q; double* pose_t; double* pt;   // estimates (all vertices)
    double* edge_blk;             // [nE][BA_EDGE_STRIDE]
    double* Hpp; double* Hll; double* b; double* x;
    double* Dinv; double* db;
    double* S;                    // dense reduced camera system, sp x sp
};

void ba_launch_error(const CorbBADev& d, double* partial, int nparts, double* out, hipStream_t s);
void ba_launch_build(const CorbBADev& d, double* maxdiag_out, hipStream_t s);
void ba_launch_schur(const CorbBADev& d, double lambda, int* bad, hipStream_t s);
void ba_launch_backsub_update(const CorbBADev& d, double lambda, double* partial, int nparts, double* scale_out, hipStream_t s);
