// ba_internal.h -- device argument block of the bundle-adjustment kernels.
#pragma once
#include "corb_internal.h"

#define BA_EDGE_STRIDE 30     // doubles per edge: A'WA(6) -A'We(3) | JB = (sqrt(w) B)' (18) | r = -sqrt(w) e (3)      (B'WA, the 6x3 Hpl block, lives in hpl[e][18])

// Levenberg-Marquardt control on the device (local windows): the accept / reject decision of a trial, the lambda update and ORB-SLAM2's stop rule
// (optimization_algorithm_levenberg.cpp:104-160, Optimizer.cc's nBad rule) by a one-thread kernel behind the trial's kernels, so that the host enqueues several LM
// iterations back to back instead of waiting for every trial's chi2 (ba_lm_device: a window's call is bound by those round trips).  A trial that is not accepted stops
// the chain -- the kernels behind it return at once -- and the host loop takes over from the estimates before that trial.
#define BA_CHAIN_MAX 16       // iterations one BALMCtl can log
#define BA_LM_CHAIN 5          // iterations per chain (tools/gpu_lba_store_timeline.sh, CORB_BA_CHAIN: see ba_lm_device)
struct BALMCtl {
    double lambda, ni, currentChi;
    double chi0;                            // begin chains: chi2 of the estimates the optimize() call started from
    int nBad, it_done, trials, stop;        // stop: 0 running / ran to its end, 2 the stop rule fired (nBad >= 3), 3 a trial was not accepted
    int iterations, begin;                  // begin: the chain starts an optimize() call -- ba_lm_begin_kernel takes lambda (computeLambdaInit) and currentChi from the scalars
    double chi2_hist[BA_CHAIN_MAX], lambda_hist[BA_CHAIN_MAX];
};
// The classification between / after the optimize() calls of a staged solve (corb_ba_solve_staged) on the device: per edge of the session's graph the chi2 of its last
// computeError() as an ACTIVE edge (last), the test of the stage (CorbBAStage), the next active set and the weights the next optimize() sees (w0 or 0).
struct BAStageDev {
    double* last; const double* w0; double* e_w;
    const unsigned char* act_in; unsigned char* act_out;
    float th_mono, th_stereo; double thd_mono, thd_stereo;
    int check_depth, recompute_inactive, allow_reactivate, float_compare;
};
// an edge as the keyframe-ordered pass needs it (observation, information weight, point vertex, dimension): 40 bytes, read as one contiguous stream per keyframe
struct BAKfRec { double obs[3]; double w; int vpoint; int dim; };
struct CorbBADev {
    int nE, nP, nL, sp;           // active edges, free poses, free landmarks, 6*nP
    int robust;
    double delta2, delta3;
    const double* cam;            // [all pose vertices][5] fx, fy, cx, cy, bf of the observing keyframe (e->fx = pKF->fx, Optimizer.cc:160-163, 189-193)
    // edges, sorted by landmark (free-pose edges first inside a landmark); edges of fixed landmarks last
    const int* e_pose; const int* e_point;        // hessian indices (-1 = fixed vertex)
    const int* e_vpose; const int* e_vpoint;      // vertex indices
    const double* e_obs; const double* e_w; const unsigned char* e_dim;
    const int* loff;              // [nL+1] edge range of each free landmark
    const int* lnfree;            // [nL]   number of free-pose edges (the leading ones)
    const int* poff; const int* pedge;            // per free pose: edge ids
    const int* pose_vertex; const int* point_vertex;
    double* pose_q; double* pose_t; double* pt;   // estimates (all vertices)
    double* edge_blk;             // [nE][edge_stride]: lean form (multi-kernel path) JB (18) | r (3); the one-workgroup optimiser keeps the 30-double records
    int edge_stride, edge_jb;     // doubles per edge record; offset of JB inside it (21 / 0 lean, 30 / 9 one-workgroup)
    int nfree_edges;              // = loff[nL]: edges of free landmarks (the rest, at the end, belong to fixed landmarks)
    int lean;                     // 1: the multi-kernel path -- no hpl array, Hll / b_l summed by the landmark's own thread while it linearises, V, the reduced
                                  //    right-hand side and the back substitution on C_l (L^-T of Hll + lambda I) and g_l = C_l' b_l instead of Dinv / db
    int v_kf, n_list;             // 1: the V blocks (bd) lie in keyframe-list order -- block i belongs to list entry pedge[i], i < n_list = poff[nP] -- instead of edge order (round 6: ba_v_kf_kernel)
    const int* vslot;             // v_kf: [nE] list position of every edge (-1: in no free keyframe's list)
    int backsub_rederive;         // lean form: the back substitution re-derives W_e' x_p from the estimates instead of reading the V blocks (ba_backsub_lean_one)
    double* hpl;                  // [nE][18] B'WA of every edge (6 x 3, row-major)
    double* e_chi2;               // [nE] chi2 of the edge's last computeError() (g2o keeps _error until the next call)
    double* Hpp; double* Hll; double* b; double* x;
    double* Dinv; double* db;      // lean: Dinv[l][0..5] = C_l (c00 c01 c02 c11 c12 c22), db[l] = g_l
    double* S;                    // dense reduced camera system, sp x sp   (solver 1)
    // block-sparse reduced camera system (solver 2): BSR with 6x6 blocks, pattern = pose pairs sharing a landmark
    const int* bsr_rowptr; const int* bsr_col; const int* bsr_diag;   // [nP+1], [nnzb], [nP] slot of (k,k)
    double* bsr_val;              // [nnzb][36]
    const int* bsr_tslot;         // [nnzb] slot the SpMV reads block s from: s itself on / above the diagonal, the transposed block's slot below it
    double* Minv;                 // [nP][36] inverse of the diagonal blocks (block-Jacobi preconditioner, pc_g == 1)
    // block-Jacobi with blocks of pc_g consecutive poses (pc_gb = 6 pc_g rows, a multiple of BA_PC_ROWS): the dense diagonal blocks of S
    // are inverted per LM trial (ba_pc_invert_kernel: one workgroup per block, in registers: ba_pc_sweep_body) and applied as dense symmetric mat-vecs inside the CG step
    int pc_g, pc_gb, pc_nblk;
    double* pc_inv;               // [pc_nblk][pc_gb][pc_gb]
    float* pc_inv32;              // the same in single precision (48 x 48 or 96 x 96 blocks): half the bytes of the largest array a CG iteration reads; NULL = pc_inv
    float* pc_pack32;             // the same blocks as their upper triangle of 16 x 16 tiles, tile-major (pc_gb / 16 = nt: nt (nt + 1) / 2 tiles of 256 floats per block): what the CG step
                                  // reads -- 58 % of the square's bytes, every off-diagonal tile serving its rows AND its columns (ba_pcg_step_sym_body); NULL: the square form
    int pc_split;                 // workgroups per block of the CG step / init kernels: 1 with pc_pack32, pc_gb / BA_PC_ROWS without
    int* pc_info;                 // [2][pc_nblk] (unused since the blocks are inverted by the library's own kernel; kept for the layout)
    double* cg_r[2]; double* cg_z; double* cg_q; double* cg_p[2];
    int cg_nparts;                // workgroups of the row-parallel CG kernels = ceil(sp/256)
    int cg_nparts_spmv;           // workgroups of the SpMV kernel (one wavefront per block row) = ceil(nP/4) rounded up to a multiple of 8 (XCD-aware row order)
    double* cg_part;              // r.z[2][cg_nparts] | r.r[2][cg_nparts] | p.q[cg_nparts_spmv]  (r.z / r.r double-buffered by parity)
    int cg_two_level;             // large systems: the partials are summed per group of 64 workgroups by the group's last workgroup (cg_part2); consumers sum the groups
    int cg_ngrp, cg_ngrp_spmv;    // groups of the vector kernels' / the SpMV's workgroups
    double* cg_part2;             // r.z[2][ngrp] | r.r[2][ngrp] | p.q[ngrp_spmv]
    int* cg_tick;                 // [ngrp + ngrp_spmv + 2] tickets (zero between kernels): one per group, then the two third-level tickets
    double* cg_fin;               // [8] final sums: r.z[2] | r.r[2] | p.q (written by the last group's wavefront, read by the next kernel)
    int* red_tick;                // ticket of the chi2 / computeScale sums (zero between kernels)
    double* cg_scal;              // [8] rz_old, rz_new, bb, pq, ...
    int* cg_flag;                 // [2] done, fail
    int use_bsr;
    int bsr_max_row;              // largest number of blocks in one block row
    // deterministic MFMA Schur: per block (p, q >= p) of the pattern the list of edge pairs (e1 = (p, l), e2 = (q, l)) over the landmarks l both
    // poses observe, ascending in l; S(p,q) = sum over the list of BD_e1 W_e2' is then ONE contraction of depth 3 x pairs per block
    int nnzb;
    int use_pairs;                // pair lists present (every multi-kernel call)
    const int* plm;               // [poff[nP]] landmark (hessian index, -1 = fixed) of every entry of pedge: ascending per pose, the -1s last
    int nu;                       // blocks on / above the diagonal
    int4* uinfo;                  // [nu] (slot, p, q, slot of the transposed block) of the u-th such block, in slot order
    int* pair_off;                // [nu + 1]
    const int2* pairs;            // [pair_off[nu]]
    double* bd;                   // [nE][18] V_e = W_e C_l (6 x 3) of the current trial, C_l C_l' = (Hll + lambda I)^-1
    // row-owner Schur kernel (ba_schur_row_kernel: block-sparse maps): a workgroup per keyframe holds the keyframe's own V blocks in LDS
    int hpp_scratch;              // 1 (maps): Hpp | b_p from kfrec -- the edges' static data in keyframe-list order -- and the estimates; the build kernel writes no JB | r records
    const struct BAKfRec* kfrec;  // [poff[nP]]
    int row_schur;                // 1: pairs[].x is the position of edge 1 in its keyframe's list (pedge[poff[p] + x]) instead of the edge id
    int* urow;                    // [nP + 1] first block (index into uinfo) of every block row
    // work decomposition of the row-owner kernel (see ba_rr_units_kernel): workgroups = (keyframe, range of its observation list)
    int n_wg, n_units, n_wb;      // workgroups, work units, entries of wb_unit
    int* rr_off; int* rowwb;      // [nP + 1] first workgroup of a keyframe; first entry of the keyframe's workgroups in wb_unit
    int4* wghdr; int* wb_off;     // [n_wg] header, first entry in wb_unit
    int* wb_unit;                 // per (workgroup, block of the row): first unit
    int* scan_scratch;            // corb_launch_exclusive_scan's scratch for the longest structure scan (pair_off, wb_unit)
    int4* units;                  // [n_units] (first pair, pairs, first list entry of the range, -)
    int* wave_off;                // [n_wg BA_ROW_WAVES + 1] first round of every wavefront of the row kernel in row_stream (round 6; NULL: the per-unit kernel)
    int2* row_stream;             // [rounds][16] the wavefronts' padded pair streams (ba_rr_stream_kernel)
    int* wunit; int4* wave_ucnt;  // [n_units] a workgroup's units grouped by wavefront, in stream order; [n_wg] units per wavefront (ba_rr_assign_kernel)
    double* upart;                // [n_units][36] partial blocks
    double* rpart;                // [n_wg][BA_ROW_WAVES][6] reduced right-hand side: a wavefront's sum of V_e g_l over its observations of the range
    const struct BAMLDev* ml;     // multilevel preconditioner (host pointer; NULL = block Jacobi only): see ba_multilevel.h
    BALMCtl* ctl;                 // device-side LM control of the running chain (NULL: the host decides; see BALMCtl)
    int row_abl;                  // -DCORB_DEV builds: timing experiments of ba_schur_row_kernel (0 = off)
    long long* row_dbg;           // -DCORB_DEV builds: per wavefront 8 cycle stamps of ba_schur_row_kernel (NULL = off)
};


// ctl_bad != nullptr (chains of LM iterations, d.ctl set): the sum is a trial's chi2 and the launch also takes the trial's decision (ba_lm_decide: out = the trial's scalars,
// ctl_bad = the two status words, ctl_epoch = the trial's number)
void ba_launch_error(const CorbBADev& d, double* partial, int nparts, double* out, hipStream_t s, const int* ctl_bad = nullptr, int ctl_epoch = 0);
void ba_launch_build(const CorbBADev& d, double* maxdiag_out, hipStream_t s, double* chi_partial = nullptr, double* chi_out = nullptr, const int* ctl_bad = nullptr, int ctl_epoch = 0);
int ba_build_lean_blocks(const CorbBADev& d);
void ba_launch_kfrec(const CorbBADev& d, BAKfRec* out, int n, hipStream_t s);
void ba_launch_schur(const CorbBADev& d, double lambda, int* bad, int epoch, int zero_S, hipStream_t s);
void ba_launch_backsub_update(const CorbBADev& d, double lambda, double* partial, int nparts, double* scale_out, double* state, double* bak, size_t n_state, hipStream_t s);

#ifndef BA_ROW_WAVES
#define BA_ROW_WAVES 4         // wavefronts of a row workgroup of ba_schur_row_kernel (a work unit per wavefront and turn); see BA_ROW_RANGE
#endif
#define BA_PC_ROWS 48         // rows of a preconditioner block handled by one workgroup of the CG step
int ba_launch_schur_bsr(const CorbBADev& d, double lambda, int nnzb, int* bad, int epoch, hipStream_t s, int pc_refresh);
int ba_launch_pc_refresh(const CorbBADev& d, hipStream_t s);      // the block preconditioner's set-up alone (what pc_refresh != 0 appends to the call above)
void ba_launch_pcg_init(const CorbBADev& d, double tol, hipStream_t s);
void ba_launch_tslot(const CorbBADev& d, int* tslot, hipStream_t s);
void ba_launch_pcg_chunk(const CorbBADev& d, int n_iter, hipStream_t s, int par0 = 0);
void ba_launch_pcg_resume(const CorbBADev& d, double tol, hipStream_t s);
// self-certification (ba_kernels.hip): true residual of the solve in d.x against the right-hand side b (part: 2 x ceil(sp / 256) doubles; out[0] max, out[1] last); |v|_inf
void ba_launch_true_residual(const CorbBADev& d, const double* b, double* part, double* out, hipStream_t s);
void ba_launch_absmax(const double* v, size_t n, double* out, hipStream_t s);
#define BA_FUSED_UPDATE_BLOCKS 1024   // workgroups up to which the oplus kernel also backs up the estimates and sums computeScale (one ticket)
#define BA_SMALL_SP 96            // dense reduced systems up to this size (16 free poses) ...
#define BA_SMALL_EDGES 2048       // ... and up to this many observations run in the fused one-workgroup optimiser (measured crossover with the multi-kernel path: 1 500 - 3 000)
struct CorbBASmall {
    int iterations;
    double* state; double* state_bak; size_t n_state;   // pose_q | pose_t | pt as one block, and its push() copy
    double* chi2_hist; double* lambda_hist;             // [iterations + 1], [iterations]
    int* counters;                                      // iterations done, trials
};
void ba_launch_small_optimize(const CorbBADev& d, const CorbBASmall& a, hipStream_t s);
// the trial's decision (see BALMCtl): scal = the trial's scalars (chi2, -, scale, ...), bad = the two status words, epoch = the trial's number
void ba_launch_lm_ctl(const CorbBADev& d, const double* scal, const int* bad, int epoch, hipStream_t s);
void ba_launch_lm_begin(const CorbBADev& d, const double* scal, hipStream_t s);      // scal[0] = chi2 of the start estimates, scal[1] = the largest diagonal entry
void ba_launch_stage_classify(const CorbBADev& d, const BAStageDev& a, hipStream_t s);      // d.e_w = the information weights w0 (the evaluation is the edge's own, not the masked one)
void ba_launch_edge_eval(const CorbBADev& d, double* chi2, double* depth, hipStream_t s);
// structure of the pair lists: count per slot + mirror slots, exclusive scan (pair_off[nnzb] = total), fill
void ba_launch_small_solve(const CorbBADev& d, int* info, hipStream_t s);      // dense reduced system with sp <= 128: one workgroup, in LDS
void ba_launch_pairs_count(const CorbBADev& d, hipStream_t s);
void ba_launch_pairs_fill(const CorbBADev& d, hipStream_t s);
void ba_launch_row_structure(const CorbBADev& d, hipStream_t s);                  // urow[] (before the pair lists)
void ba_launch_rr_count(const CorbBADev& d, hipStream_t s);                        // ranges per keyframe (scanned)
void ba_launch_rr_units(const CorbBADev& d, bool fill, hipStream_t s);             // units per workgroup (count + scan), then the tables
void ba_launch_vslot(const CorbBADev& d, int* vslot, int n_edges, int n_list, hipStream_t s);      // keyframe-list order of the V blocks: edge -> list position
void ba_launch_rr_stream(const CorbBADev& d, bool fill, hipStream_t s);            // rounds per wavefront (count; the caller scans wave_off), then the padded streams
#define BA_ROW_MIN_POSES 64        // block-sparse maps from this many free keyframes on run the row-owner Schur kernel
