// ba_multilevel.h -- additive multilevel preconditioner of the reduced camera system (solver 2, large maps).
// The block-Jacobi preconditioner (16-keyframe blocks) leaves the smooth error along a trajectory -- drift -- to the Krylov iteration: the CG iterations
// of one solve grow from 17 (first LM iteration, large lambda) to 800+ (tenth) on a 6 250-keyframe loop (tools/pcg_proto.py on reduced systems dumped by
// the oracle).  Here the keyframes, in index order, carry a hierarchy of coarse "nodes" with LINEAR hat interpolation per 6-dof component (stride 8, then 4
// per level, until 16 nodes are left): z = D0^-1 r + sum_k W_k' D_k^-1 W_k r with W_k the composite restriction to level k, A_k = W_k S W_k' the Galerkin
// matrices (block-sparse, built level by level) and D_k their block-Jacobi blocks (16 nodes; the top level is one block, i.e. exact).  A BPX-type
// preconditioner: symmetric positive definite by construction, every sum in a fixed order.  Prototype: 818 -> 64 iterations on the hardest system, 51 -> 28
// on an easy one; piecewise-constant aggregation instead of hats gave 226.
#pragma once
#include "ba_internal.h"

#define BA_ML_MAX_LEVELS 10
#define BA_ML_AUTO_POSES 256    // free keyframes from which the coarse levels are used by default (CorbBAOptions.pc_multilevel): measured per 10 LM iterations, without / with them,
                                // 600 keyframes 26.7 / 25.7 ms, 1 200: 33.1 / 25.7, 2 x 800: 37.6 / 34.5, 280 keyframes 20.0 / 10.7 ms, 400: 23.4 / 10.4, 600: 26.7 / 12.1, 1 200: 32.7 / 12.9, 2 000: 35.0 / 14.3 (tools/ml_small.py; 2 048 until late in round 4):
                                // every map the PCG solver takes
#define BA_ML_STRIDE0 8          // keyframes per node of the first coarse level (the deeper levels: 4 nodes per node)
#define BA_ML_WEIGHT 1.0         // default weight of the coarse levels' terms (corb_ba.cpp ml_level_weight)
#define BA_ML_CHUNK 128          // entries of a restriction row summed by one wavefront
#define BA_ML_G 16              // nodes per block-Jacobi block of a coarse level (96 rows: 16 x 16 threads with a 6 x 6 block each in ba_pc_sweep_body)
struct BAMLLevel {
    int n, nblk;                // nodes, block-Jacobi blocks
    int stride;                 // coarsening factor from the level below (level 0 = the keyframes)
    int nnzb, max_row;          // blocks of A_k, most blocks in one of its rows
    int node_off, blk_off;      // first node / first block of the level in the all-level arrays
    int* rowptr; int* col; double* val;       // A_k (BSR, 6 x 6 blocks)
    float* pc_inv32;            // [nblk][96][96]
    double wgt;                 // weight of the level's term W_k' D_k^-1 W_k in the additive sum (ml_level_weight)
    // hats of this level over the nodes of the level below (tables: the hats do not cross trajectory boundaries): node i below <- nodes i0[i], i1[i] with
    // weights 1 - w1[i], w1[i]; the nodes below under the hat of node I are lo[I] .. hi[I]
    const int* i0; const int* i1; const double* w1; const int* lo; const int* hi;
};
struct BAMLDev {
    int L, n_nodes, n_blocks;
    BAMLLevel lv[BA_ML_MAX_LEVELS];
    const int* r_ptr; const int* r_pose; const double* r_w;    // node g <- (keyframe, weight): composite restriction W_k, rows of all levels
    // the rows are summed in chunks of at most BA_ML_CHUNK entries, a wavefront each (a top-level node gathers from thousands of keyframes): chunk c covers
    // entries ch_begin[c] .. ch_begin[c + 1] of its node; node g owns the chunks ch_ptr[g] .. ch_ptr[g + 1], whose sums the block kernel adds in order
    int n_chunks; const int* ch_begin; const int* ch_ptr; double* ch_sum;      // ch_sum: [n_chunks][6]
    const int* p_ptr; const int* p_node; const double* p_w;    // keyframe <- (node, weight): its transpose
    double* rk; double* yk;     // [6 n_nodes] restricted residuals / corrections
    int np, ngrp;               // workgroups of the prolongation kernel, groups of 64 of them
    double* part; double* part2; int* tick;                    // r.z partial sums of the prolongation kernel (cg_tree_reduce)
};
// per LM trial (after S): A_1 .. A_L by Galerkin products, their diagonal blocks inverted
void ba_ml_launch_setup(const CorbBADev& d, const BAMLDev& m, hipStream_t s);
// z += coarse corrections of r (= d.cg_r[r_buf]); r.z of the full preconditioner into the final slot of parity `par` (both parities: init)
void ba_ml_launch_apply(const CorbBADev& d, const BAMLDev& m, int r_buf, int par, int both, hipStream_t s);
void ba_ml_launch_prolong(const CorbBADev& d, const BAMLDev& m, int par, hipStream_t s);
void ba_ml_launch_step_coarse(const CorbBADev& d, const BAMLDev& m, int par, hipStream_t s);
