/* CPU ORACLE (test infrastructure) -- global bundle adjustment.
 * Restates Optimizer::BundleAdjustment (corbslam_client/src/Optimizer.cc:54-270) and the g2o pieces it
 * drives (G/ = corbslam_client/Thirdparty/g2o/g2o/):
 *   G/types/types_six_dof_expmap.{h,cpp}  EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ error + Jacobians
 *   G/types/se3quat.h, se3_ops.hpp        SE3Quat exp / product / map
 *   G/core/base_binary_edge.hpp:55-120    constructQuadraticForm (robust and non-robust branch)
 *   G/core/block_solver.hpp:354-604       Schur complement, setLambda / restoreDiagonal
 *   G/core/optimization_algorithm_levenberg.cpp:61-189   LM control flow, lambda init, scale
 *   G/core/sparse_optimizer.cpp:100-114, 354-419         activeRobustChi2, optimize loop
 *   G/core/robust_kernel_impl.cpp:78-91   Huber
 * Eigen (absent here) is replaced by plain loops: Quaterniond(R), q*v, q*q, toRotationMatrix and 3x3
 * inverse follow Eigen 3's formulas; the reduced system is solved by a dense LDL^T without pivoting
 * (the reference: Eigen SimplicialLDLT with AMD ordering -- same factorisation up to rounding order).
 * Tolerance of the parity claim for this part: 1e-4 relative (BASELINE.json north_star), FP64 inside,
 * FP32 at the boundary (Converter.cc:37-92).  See orc.h for scope. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>

typedef struct { double q[4]; /* x y z w */ double t[3]; } SE3;

static void quat_normalize(double* q)
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }      /* normalizeRotation (se3quat.h:289-294) */
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl) */
static void quat_from_R(const double R[9], double* q)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const double* q, double R[9])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* Eigen QuaternionBase::_transformVector */
static void quat_rot(const double* q, const double* v, double* o)
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

static void quat_mul(const double* a, const double* b, double* o)      /* a * b */
{
    double r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    memcpy(o, r, sizeof(r));
}

/* SE3Quat::exp (se3quat.h:223-257): update = (omega, upsilon) */
static void se3_exp(const double* u, SE3* out)
{
    const double* om = u; const double* up = u + 3;
    double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += O[i * 3 + k] * O[k * 3 + j]; O2[i * 3 + j] = s; }
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (pow(theta, 3));
        for (int i = 0; i < 9; i++) {
            double I = (i % 4) == 0 ? 1.0 : 0.0;
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    quat_from_R(R, out->q);
    for (int i = 0; i < 3; i++) out->t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    quat_normalize(out->q);
}

/* SE3Quat::operator* (se3quat.h:102-108) */
static void se3_mul(const SE3* a, const SE3* b, SE3* o)
{
    SE3 r; double rt[3];
    quat_rot(a->q, b->t, rt);
    r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
    quat_mul(a->q, b->q, r.q);
    quat_normalize(r.q);
    *o = r;
}

typedef struct {
    int pose, point;        /* hessian indices or -1 if fixed */
    int vpose, vpoint;      /* vertex indices */
    int dim;                /* 2 mono, 3 stereo */
    double obs[3], w;       /* information = w * I */
    int orig;               /* index in the caller's edge array */
} Edge;

typedef struct {
    int K, M, E;
    SE3* pose; double* pt;
    const OrcBAProblem* prob;
    Edge* e;
    int nP, nL;                 /* free poses / free points */
    const double* cam;          /* [K][5] fx, fy, cx, cy, bf of every pose vertex (e->fx = pKF->fx ... e->bf = pKF->mbf, Optimizer.cc:160-163, 189-193) */
    int robust;
    double d2, d3;              /* Huber deltas (mono / stereo edges) */
    double* last_chi2;          /* per ORIGINAL edge: chi2 of the last computeError() on it (g2o keeps _error stale) */
} BA;

static double edge_error(const BA* ba, const Edge* e, double* err)
{
    double Xc[3];
    quat_rot(ba->pose[e->vpose].q, ba->pt + 3 * e->vpoint, Xc);
    Xc[0] += ba->pose[e->vpose].t[0]; Xc[1] += ba->pose[e->vpose].t[1]; Xc[2] += ba->pose[e->vpose].t[2];
    const double* cam = ba->cam + 5 * (size_t)e->vpose;
    if (e->dim == 2) {                                   /* cam_project (types_six_dof_expmap.cpp:141-147) */
        err[0] = e->obs[0] - (Xc[0] / Xc[2] * cam[0] + cam[2]);
        err[1] = e->obs[1] - (Xc[1] / Xc[2] * cam[1] + cam[3]);
        err[2] = 0;
        return e->w * (err[0] * err[0] + err[1] * err[1]);
    }
    const float invz = (float)(1.0f / Xc[2]);            /* :151  `const float invz = 1.0f/trans_xyz[2]` */
    double r0 = Xc[0] * invz * cam[0] + cam[2];
    double r1 = Xc[1] * invz * cam[1] + cam[3];
    double r2 = r0 - cam[4] * invz;
    err[0] = e->obs[0] - r0; err[1] = e->obs[1] - r1; err[2] = e->obs[2] - r2;
    return e->w * (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]);
}

static void huber(double e, double delta, double* rho)   /* robust_kernel_impl.cpp:78-91 */
{
    double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}

static double active_robust_chi2(const BA* ba)
{
    const double d2 = ba->d2, d3 = ba->d3;
    double chi = 0, err[3], rho[3];
    for (int i = 0; i < ba->E; i++) {
        double c = edge_error(ba, &ba->e[i], err);
        if (ba->last_chi2) ba->last_chi2[ba->e[i].orig] = c;
        if (ba->robust) { huber(c, ba->e[i].dim == 2 ? d2 : d3, rho); chi += rho[0]; }
        else chi += c;
    }
    return chi;
}

/* linearizeOplus (types_six_dof_expmap.cpp:103-139, 188-234): A = d e / d point (dim x 3), B = d e / d pose (dim x 6) */
static void edge_jacobians(const BA* ba, const Edge* e, double* A, double* B)
{
    const SE3* T = &ba->pose[e->vpose];
    double R[9]; quat_to_R(T->q, R);
    double Xc[3]; quat_rot(T->q, ba->pt + 3 * e->vpoint, Xc);
    Xc[0] += T->t[0]; Xc[1] += T->t[1]; Xc[2] += T->t[2];
    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
    const double* cam = ba->cam + 5 * (size_t)e->vpose;
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    if (e->dim == 2) {
        double tmp[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) {
            double s = 0; for (int k = 0; k < 3; k++) s += tmp[i * 3 + k] * R[k * 3 + j];
            A[i * 3 + j] = -1. / z * s;
        }
    } else {
        for (int j = 0; j < 3; j++) {
            A[0 * 3 + j] = -fx * R[0 * 3 + j] / z + fx * x * R[2 * 3 + j] / z_2;
            A[1 * 3 + j] = -fy * R[1 * 3 + j] / z + fy * y * R[2 * 3 + j] / z_2;
            A[2 * 3 + j] = A[0 * 3 + j] - bf * R[2 * 3 + j] / z_2;
        }
    }
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (e->dim == 3) {
        B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2;
    }
}

static int inv3(const double* m, double* o)          /* Eigen Matrix3d::inverse(): cofactors / determinant */
{
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return isfinite(id);
}

/* dense LDL^T (no pivoting) of the symmetric n x n matrix a (row-major, full), solves a x = b in place */
static int ldlt_solve(double* a, int n, double* b)
{
    for (int j = 0; j < n; j++) {
        double d = a[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k] * a[(size_t)k * n + k];
        if (!isfinite(d) || d == 0.0) return 0;
        a[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = a[(size_t)i * n + j];
            const double* ri = a + (size_t)i * n; const double* rj = a + (size_t)j * n;
            for (int k = 0; k < j; k++) s -= ri[k] * rj[k] * a[(size_t)k * n + k];
            a[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= a[(size_t)i * n + k] * b[k]; b[i] = s; }
    for (int i = 0; i < n; i++) b[i] /= a[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= a[(size_t)k * n + i] * b[k]; b[i] = s; }
    return 1;
}

/* optimizer.initializeOptimization(level 0) + optimize(iters) on the edges with active[i] != 0, starting from and
 * updating the double-precision estimates `pose` / `pt`.  chi2 / lambda histories are optional. */

/* ------------------------------------------------------------------------------------------------------------------------------
 * Block-sparse reduced camera system and its LDL^T -- the solver CLASS of the reference: BlockSolver keeps Hschur as a sparse block matrix
 * whose pattern is the pose pairs that share a landmark (block_solver.hpp:262-292), LinearSolverEigen copies its upper triangle into an
 * Eigen::SparseMatrix and factorises it with SimplicialLDLT after a fill-reducing ordering computed once (linear_solver_eigen.h:94-232).
 * Eigen is absent: here the matrix is block-CSR (6x6 blocks, upper triangle), the ordering is reverse Cuthill-McKee on the block graph
 * (Eigen: AMD on the scalar graph -- any symmetric permutation gives the same solution up to rounding), and the factorisation is an
 * up-looking block LDL^T along the elimination tree (the algorithm SimplicialLDLT implements, on 6x6 blocks).  Used above ORC_BA_DENSE_MAX
 * free poses, where the dense n^3 factorisation stops being a usable CPU baseline; tests compare both solvers on the same problems. */
#define ORC_BA_DENSE_MAX 256
static int g_force_solver = 0;                 /* 0 auto, 1 dense, 2 sparse */
void orc_ba_set_solver(int solver) { g_force_solver = solver; }

typedef struct {
    int n;                      /* free poses */
    int* perm; int* iperm;      /* perm[new] = old, iperm[old] = new */
    int* rowptr; int* col;      /* upper triangle incl. diagonal, permuted indices, columns ascending */
    double* val;                /* [nnz][36] */
    int* cptr; int* crow; int* cslot;     /* column lists of the strict upper triangle: for column k the rows i < k and their slots */
    int* parent;                /* elimination tree */
    int* lptr; int* lrow; double* lval;   /* L by columns (unit diagonal not stored), rows ascending */
    double* D; double* Dinv;    /* [n][36] */
    int* flag; int* stack; double* W;     /* work: n, n, [n][36] */
    int* lnext;
    long nnz_l;
} BSys;

static int cmp_ll(const void* a, const void* b) { long long x = *(const long long*)a, y = *(const long long*)b; return x < y ? -1 : x > y; }

static void bsys_free(BSys* B)
{
    free(B->perm); free(B->iperm); free(B->rowptr); free(B->col); free(B->val); free(B->cptr); free(B->crow); free(B->cslot); free(B->parent);
    free(B->lptr); free(B->lrow); free(B->lval); free(B->D); free(B->Dinv); free(B->flag); free(B->stack); free(B->W); free(B->lnext);
}

/* pattern from the landmarks' free-pose edge lists, RCM ordering, symbolic factorisation */
static void bsys_build(BSys* B, int nP, int nL, const int* loff, const int* ledge, const Edge* e)
{
    memset(B, 0, sizeof(*B)); B->n = nP;
    /* unique off-diagonal pairs (p < q) */
    size_t cap = 0;
    for (int l = 0; l < nL; l++) { size_t k = loff[l + 1] - loff[l]; cap += k * (k - 1) / 2; }
    long long* key = (long long*)malloc(sizeof(long long) * (cap ? cap : 1)); size_t nk = 0;
    for (int l = 0; l < nL; l++)
        for (int a = loff[l]; a < loff[l + 1]; a++) for (int b = a + 1; b < loff[l + 1]; b++) {
            int p = e[ledge[a]].pose, q = e[ledge[b]].pose;
            if (p < 0 || q < 0 || p == q) continue;
            if (p > q) { int t = p; p = q; q = t; }
            key[nk++] = (long long)p * nP + q;
        }
    qsort(key, nk, sizeof(long long), cmp_ll);
    size_t nu = 0; for (size_t i = 0; i < nk; i++) if (i == 0 || key[i] != key[i - 1]) key[nu++] = key[i];
    /* symmetric adjacency */
    int* deg = (int*)calloc(nP + 1, sizeof(int));
    for (size_t i = 0; i < nu; i++) { deg[key[i] / nP]++; deg[key[i] % nP]++; }
    int* aptr = (int*)malloc(sizeof(int) * (nP + 1)); aptr[0] = 0; for (int i = 0; i < nP; i++) aptr[i + 1] = aptr[i] + deg[i];
    int* adj = (int*)malloc(sizeof(int) * (aptr[nP] ? aptr[nP] : 1)); int* cur = (int*)malloc(sizeof(int) * (nP + 1)); memcpy(cur, aptr, sizeof(int) * (nP + 1));
    for (size_t i = 0; i < nu; i++) { int p = (int)(key[i] / nP), q = (int)(key[i] % nP); adj[cur[p]++] = q; adj[cur[q]++] = p; }
    /* reverse Cuthill-McKee: BFS from a pseudo-peripheral vertex of every component, neighbours by ascending degree */
    B->perm = (int*)malloc(sizeof(int) * (nP ? nP : 1)); B->iperm = (int*)malloc(sizeof(int) * (nP ? nP : 1));
    int* order = (int*)malloc(sizeof(int) * (nP ? nP : 1)); int* lvl = (int*)malloc(sizeof(int) * (nP ? nP : 1)); int no = 0;
    char* seen = (char*)calloc(nP ? nP : 1, 1);
    for (int s0 = 0; s0 < nP; s0++) {
        if (seen[s0]) continue;
        int start = s0;
        for (int rep = 0; rep < 3; rep++) {                     /* walk to the far end of the component a few times */
            int h = 0, t = 0; order[no + t++] = start; lvl[start] = 0;
            char* m = (char*)calloc(nP, 1); m[start] = 1;
            while (h < t) { int v = order[no + h++]; for (int a = aptr[v]; a < aptr[v + 1]; a++) { int w = adj[a]; if (!m[w] && !seen[w]) { m[w] = 1; lvl[w] = lvl[v] + 1; order[no + t++] = w; } } }
            int far = order[no + t - 1];
            for (int i = t - 1; i >= 0 && lvl[order[no + i]] == lvl[far]; i--) if (deg[order[no + i]] < deg[far]) far = order[no + i];
            free(m);
            if (far == start) break;
            start = far;
        }
        int h = no, t = no; order[t++] = start; seen[start] = 1;
        while (h < t) {
            int v = order[h++]; int t0 = t;
            for (int a = aptr[v]; a < aptr[v + 1]; a++) { int w = adj[a]; if (!seen[w]) { seen[w] = 1; order[t++] = w; } }
            for (int i = t0 + 1; i < t; i++) { int w = order[i], j = i; while (j > t0 && deg[order[j - 1]] > deg[w]) { order[j] = order[j - 1]; j--; } order[j] = w; }   /* insertion sort by degree */
        }
        no = t;
    }
    for (int i = 0; i < nP; i++) { B->perm[i] = order[nP - 1 - i]; B->iperm[B->perm[i]] = i; }
    free(order); free(lvl); free(seen); free(adj); free(aptr); free(cur); free(deg);
    /* upper block-CSR in the permuted numbering */
    for (size_t i = 0; i < nu; i++) { int p = B->iperm[key[i] / nP], q = B->iperm[key[i] % nP]; if (p > q) { int t = p; p = q; q = t; } key[i] = (long long)p * nP + q; }
    qsort(key, nu, sizeof(long long), cmp_ll);
    B->rowptr = (int*)calloc(nP + 2, sizeof(int)); B->col = (int*)malloc(sizeof(int) * (nu + nP + 1));
    { size_t i = 0; int nz = 0;
      for (int p = 0; p < nP; p++) { B->rowptr[p] = nz; B->col[nz++] = p; while (i < nu && key[i] / nP == p) B->col[nz++] = (int)(key[i++] % nP); }
      B->rowptr[nP] = nz; }
    free(key);
    const int nnz = B->rowptr[nP];
    B->val = (double*)malloc(sizeof(double) * 36 * (size_t)(nnz ? nnz : 1));
    /* column lists of the strict upper part */
    B->cptr = (int*)calloc(nP + 2, sizeof(int));
    for (int p = 0; p < nP; p++) for (int s = B->rowptr[p] + 1; s < B->rowptr[p + 1]; s++) B->cptr[B->col[s] + 1]++;
    for (int k = 0; k < nP; k++) B->cptr[k + 1] += B->cptr[k];
    B->crow = (int*)malloc(sizeof(int) * (nnz ? nnz : 1)); B->cslot = (int*)malloc(sizeof(int) * (nnz ? nnz : 1));
    { int* c2 = (int*)malloc(sizeof(int) * (nP + 1)); memcpy(c2, B->cptr, sizeof(int) * (nP + 1));
      for (int p = 0; p < nP; p++) for (int s = B->rowptr[p] + 1; s < B->rowptr[p + 1]; s++) { int k = B->col[s]; B->crow[c2[k]] = p; B->cslot[c2[k]++] = s; }
      free(c2); }
    /* elimination tree (Liu), then the column counts of L by one symbolic sweep */
    B->parent = (int*)malloc(sizeof(int) * (nP ? nP : 1)); int* anc = (int*)malloc(sizeof(int) * (nP ? nP : 1));
    for (int k = 0; k < nP; k++) {
        B->parent[k] = -1; anc[k] = -1;
        for (int a = B->cptr[k]; a < B->cptr[k + 1]; a++) {
            int i = B->crow[a];
            while (i != -1 && i < k) { int nx = anc[i]; anc[i] = k; if (nx == -1) B->parent[i] = k; i = nx; }
        }
    }
    free(anc);
    B->flag = (int*)malloc(sizeof(int) * (nP ? nP : 1)); B->stack = (int*)malloc(sizeof(int) * (nP ? nP : 1));
    B->lptr = (int*)calloc(nP + 2, sizeof(int));
    for (int k = 0; k < nP; k++) B->flag[k] = -1;
    for (int k = 0; k < nP; k++) {
        B->flag[k] = k;
        for (int a = B->cptr[k]; a < B->cptr[k + 1]; a++)
            for (int i = B->crow[a]; i != -1 && B->flag[i] != k; i = B->parent[i]) { B->flag[i] = k; B->lptr[i + 1]++; }     /* row k of L has an entry in column i */
    }
    for (int k = 0; k < nP; k++) B->lptr[k + 1] += B->lptr[k];
    B->nnz_l = B->lptr[nP];
    B->lrow = (int*)malloc(sizeof(int) * (B->nnz_l ? B->nnz_l : 1)); B->lval = (double*)malloc(sizeof(double) * 36 * (size_t)(B->nnz_l ? B->nnz_l : 1));
    B->D = (double*)malloc(sizeof(double) * 36 * (size_t)(nP ? nP : 1)); B->Dinv = (double*)malloc(sizeof(double) * 36 * (size_t)(nP ? nP : 1));
    B->W = (double*)malloc(sizeof(double) * 36 * (size_t)(nP ? nP : 1)); B->lnext = (int*)malloc(sizeof(int) * (nP + 1));
}

static int bsys_slot(const BSys* B, int p, int q)             /* p <= q, permuted */
{
    int lo = B->rowptr[p], hi = B->rowptr[p + 1] - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (B->col[mid] < q) lo = mid + 1; else hi = mid; }
    return lo;
}

/* 6x6 symmetric inverse through LDL^T without pivoting; 0 if a pivot is zero / not finite */
static int inv6(const double* a, double* o)
{
    double l[36], dd[6];
    for (int j = 0; j < 6; j++) {
        double d = a[j * 6 + j];
        for (int k = 0; k < j; k++) d -= l[j * 6 + k] * l[j * 6 + k] * dd[k];
        if (!isfinite(d) || d == 0.0) return 0;
        dd[j] = d; l[j * 6 + j] = 1;
        for (int i = j + 1; i < 6; i++) { double s = a[i * 6 + j]; for (int k = 0; k < j; k++) s -= l[i * 6 + k] * l[j * 6 + k] * dd[k]; l[i * 6 + j] = s / d; }
    }
    for (int c = 0; c < 6; c++) {
        double x[6];
        for (int i = 0; i < 6; i++) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; k++) s -= l[i * 6 + k] * x[k]; x[i] = s; }
        for (int i = 0; i < 6; i++) x[i] /= dd[i];
        for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= l[k * 6 + i] * x[k]; x[i] = s; }
        for (int i = 0; i < 6; i++) o[i * 6 + c] = x[i];
    }
    return 1;
}

/* numeric up-looking block LDL^T of the current values, then solve A x = b in place (x, b in ORIGINAL numbering) */
static int bsys_factor_solve(BSys* B, double* x)
{
    const int n = B->n;
    for (int k = 0; k < n; k++) { B->flag[k] = -1; B->lnext[k] = B->lptr[k]; }
    for (int k = 0; k < n; k++) {
        /* pattern of row k of L: reach of the column's rows in the elimination tree, topological order on the stack */
        int top = n; B->flag[k] = k;
        for (int a = B->cptr[k]; a < B->cptr[k + 1]; a++) {
            int i = B->crow[a], len = 0;
            const double* Aik = B->val + 36 * (size_t)B->cslot[a];           /* block (i, k), i < k */
            double* Wi = B->W + 36 * (size_t)i;
            for (int j = i; j != -1 && B->flag[j] != k; j = B->parent[j]) { B->stack[len++] = j; B->flag[j] = k; memset(B->W + 36 * (size_t)j, 0, sizeof(double) * 36); }
            while (len > 0) B->stack[--top] = B->stack[--len];
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) Wi[r * 6 + c] += Aik[c * 6 + r];      /* W_i = Z_ki = A_ki = A_ik' */
        }
        double Dk[36]; memcpy(Dk, B->val + 36 * (size_t)B->rowptr[k], sizeof(Dk));
        for (; top < n; top++) {
            const int m = B->stack[top];
            const double* Wm = B->W + 36 * (size_t)m;                        /* Z_km, final */
            double Lkm[36];
            const double* Di = B->Dinv + 36 * (size_t)m;
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s2 = 0; for (int t = 0; t < 6; t++) s2 += Wm[r * 6 + t] * Di[t * 6 + c]; Lkm[r * 6 + c] = s2; }   /* L_km = Z_km D_m^-1 */
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s2 = 0; for (int t = 0; t < 6; t++) s2 += Wm[r * 6 + t] * Lkm[c * 6 + t]; Dk[r * 6 + c] -= s2; }  /* D_k -= Z_km L_km' */
            for (int a = B->lptr[m]; a < B->lnext[m]; a++) {                 /* rows i (m < i < k) of column m: Z_ki -= Z_km L_im' */
                const int i = B->lrow[a]; const double* Lim = B->lval + 36 * (size_t)a; double* Wi = B->W + 36 * (size_t)i;
                for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s2 = 0; for (int t = 0; t < 6; t++) s2 += Wm[r * 6 + t] * Lim[c * 6 + t]; Wi[r * 6 + c] -= s2; }
            }
            const int pos = B->lnext[m]++; B->lrow[pos] = k; memcpy(B->lval + 36 * (size_t)pos, Lkm, sizeof(Lkm));
        }
        memcpy(B->D + 36 * (size_t)k, Dk, sizeof(Dk));
        if (!inv6(Dk, B->Dinv + 36 * (size_t)k)) return 0;
    }
    /* solve: y = P b; L z = y; z = D^-1 z; L' w = z; x = P' w */
    double* y = (double*)malloc(sizeof(double) * 6 * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < n; k++) memcpy(y + 6 * k, x + 6 * B->perm[k], sizeof(double) * 6);
    for (int j = 0; j < n; j++)
        for (int a = B->lptr[j]; a < B->lptr[j + 1]; a++) { const double* L = B->lval + 36 * (size_t)a; double* yi = y + 6 * B->lrow[a]; const double* yj = y + 6 * j;
            for (int r = 0; r < 6; r++) { double s2 = 0; for (int c = 0; c < 6; c++) s2 += L[r * 6 + c] * yj[c]; yi[r] -= s2; } }
    for (int j = 0; j < n; j++) { double t[6]; const double* Di = B->Dinv + 36 * (size_t)j; for (int r = 0; r < 6; r++) { double s2 = 0; for (int c = 0; c < 6; c++) s2 += Di[r * 6 + c] * y[6 * j + c]; t[r] = s2; } memcpy(y + 6 * j, t, sizeof(t)); }
    for (int j = n - 1; j >= 0; j--)
        for (int a = B->lptr[j]; a < B->lptr[j + 1]; a++) { const double* L = B->lval + 36 * (size_t)a; const double* yi = y + 6 * B->lrow[a]; double* yj = y + 6 * j;
            for (int c = 0; c < 6; c++) { double s2 = 0; for (int r = 0; r < 6; r++) s2 += L[r * 6 + c] * yi[r]; yj[c] -= s2; } }
    for (int k = 0; k < n; k++) memcpy(x + 6 * B->perm[k], y + 6 * k, sizeof(double) * 6);
    free(y);
    return 1;
}

/* development aid (tools/pcg_proto.py): the reduced system as raw arrays -- n, nnz | perm[n] | rowptr[n+1] | col[nnz] | val[nnz][36] | rhs[6n] (rhs in original numbering) */
static void bsys_dump(const BSys* B, const double* rhs, const char* prefix)
{
    static int seq = 0; char path[512]; snprintf(path, sizeof(path), "%s.%03d.bin", prefix, seq++);
    FILE* f = fopen(path, "wb"); if (!f) return;
    const int n = B->n, nnz = B->rowptr[n]; int hdr[2] = { n, nnz };
    fwrite(hdr, sizeof(int), 2, f); fwrite(B->perm, sizeof(int), n, f); fwrite(B->rowptr, sizeof(int), n + 1, f); fwrite(B->col, sizeof(int), nnz, f);
    fwrite(B->val, sizeof(double), 36 * (size_t)nnz, f); fwrite(rhs, sizeof(double), 6 * (size_t)n, f); fclose(f);
}

/* per-pose intrinsics as doubles: p->intr (n_poses x 5 floats, one row per keyframe) or the shared values */
static double* cam_table(const OrcBAProblem* p)
{
    double* cam = (double*)malloc(sizeof(double) * 5 * (p->n_poses > 0 ? p->n_poses : 1));
    for (int k = 0; k < p->n_poses; k++) {
        double* c = cam + 5 * (size_t)k;
        if (p->intr) for (int a = 0; a < 5; a++) c[a] = p->intr[5 * (size_t)k + a];
        else { c[0] = p->fx; c[1] = p->fy; c[2] = p->cx; c[3] = p->cy; c[4] = p->bf; }
    }
    return cam;
}

static int ba_optimize(const OrcBAProblem* p, const double* cam, const uint8_t* active, SE3* pose_io, double* pt_io, int iters, int robust,
                       volatile int* stop, double* chi2_hist, double* lambda_hist, int* iters_done, int* trials_done, double* last_chi2,
                       double delta2, double delta3)
{
    BA ba; memset(&ba, 0, sizeof(ba));
    ba.K = p->n_poses; ba.M = p->n_points; ba.prob = p; ba.robust = robust; ba.last_chi2 = last_chi2;
    ba.cam = cam;                                                                   /* e->fx = pKF->fx (float -> double) */
    ba.pose = pose_io; ba.pt = pt_io; ba.d2 = delta2; ba.d3 = delta3;
    int* pidx = (int*)malloc(sizeof(int) * (ba.K > 0 ? ba.K : 1));
    int* lidx = (int*)malloc(sizeof(int) * (ba.M > 0 ? ba.M : 1));
    OrcBAResult rr; memset(&rr, 0, sizeof(rr)); rr.chi2 = chi2_hist; rr.lambda = lambda_hist;
    OrcBAResult* r = &rr;
    /* active edges (allVerticesFixed dropped, sparse_optimizer.cpp:234); points without edges are removed (Optimizer.cc:198-202) */
    int* deg = (int*)calloc(ba.M > 0 ? ba.M : 1, sizeof(int));
    ba.e = (Edge*)malloc(sizeof(Edge) * (p->n_edges > 0 ? p->n_edges : 1));
    for (int i = 0; i < p->n_edges; i++) {
        const OrcBAEdge* s = &p->edges[i];
        if (s->pose < 0 || s->pose >= ba.K || s->point < 0 || s->point >= ba.M) { free(pidx); free(lidx); free(deg); free(ba.e); return -2; }
        if (active && !active[i]) continue;
        if (p->pose_fixed[s->pose] && p->point_fixed[s->point]) continue;
        Edge* e = &ba.e[ba.E++];
        e->orig = i;
        e->vpose = s->pose; e->vpoint = s->point;
        e->dim = s->ur < 0 ? 2 : 3;                                               /* mvuRight<0 -> mono edge (:147) */
        e->obs[0] = s->u; e->obs[1] = s->v; e->obs[2] = s->ur; e->w = s->inv_sigma2;
        deg[s->point]++;
    }
    /* index mapping: free poses then free points, ascending vertex id (sparse_optimizer.cpp:166-190) */
    ba.nP = 0; for (int k = 0; k < ba.K; k++) pidx[k] = p->pose_fixed[k] ? -1 : ba.nP++;
    ba.nL = 0; for (int m = 0; m < ba.M; m++) lidx[m] = (p->point_fixed[m] || deg[m] == 0) ? -1 : ba.nL++;
    for (int i = 0; i < ba.E; i++) { ba.e[i].pose = pidx[ba.e[i].vpose]; ba.e[i].point = lidx[ba.e[i].vpoint]; }
    const int nP = ba.nP, nL = ba.nL, sp = 6 * nP, sl = 3 * nL;
    /* per-landmark edge lists */
    int* loff = (int*)calloc(nL + 2, sizeof(int)); int* ledge = (int*)malloc(sizeof(int) * (ba.E > 0 ? ba.E : 1));
    for (int i = 0; i < ba.E; i++) if (ba.e[i].point >= 0) loff[ba.e[i].point + 1]++;
    for (int l = 0; l < nL; l++) loff[l + 1] += loff[l];
    { int* cur = (int*)malloc(sizeof(int) * (nL + 1)); memcpy(cur, loff, sizeof(int) * (nL + 1));
      for (int i = 0; i < ba.E; i++) { if (ba.e[i].point >= 0) ledge[cur[ba.e[i].point]++] = i; }
      free(cur); }
    double* Hpp = (double*)calloc((size_t)(nP > 0 ? nP : 1) * 36, sizeof(double));
    double* Hll = (double*)calloc((size_t)(nL > 0 ? nL : 1) * 9, sizeof(double));
    double* Hpl = (double*)calloc((size_t)(ba.E > 0 ? ba.E : 1) * 18, sizeof(double));   /* per edge: 6x3 = B^T W A */
    double* b = (double*)calloc((size_t)(sp + sl > 0 ? sp + sl : 1), sizeof(double));
    double* x = (double*)calloc((size_t)(sp + sl > 0 ? sp + sl : 1), sizeof(double));
    const int sparse = g_force_solver == 2 || (g_force_solver == 0 && nP > ORC_BA_DENSE_MAX);
    BSys bs_sys; if (sparse) bsys_build(&bs_sys, nP, nL, loff, ledge, ba.e);
    double* S = sparse ? NULL : (double*)malloc(sizeof(double) * (size_t)(sp > 0 ? sp : 1) * (sp > 0 ? sp : 1));
    double* bs = (double*)malloc(sizeof(double) * (sp > 0 ? sp : 1));
    double* Dinv = (double*)malloc(sizeof(double) * 9 * (nL > 0 ? nL : 1));
    SE3* pose_bak = (SE3*)malloc(sizeof(SE3) * (ba.K > 0 ? ba.K : 1));
    double* pt_bak = (double*)malloc(sizeof(double) * 3 * (ba.M > 0 ? ba.M : 1));
    const double d2 = delta2, d3 = delta3;
    double lambda = -1, ni = 2; int nBad = 0;
    int it_done = 0, trials_total = 0;
    if (r->chi2) r->chi2[0] = active_robust_chi2(&ba);
    int ok = 1;
    for (int it = 0; it < iters && !(stop && *stop) && ok && (nP + nL) > 0; it++) {
        /* ---- OptimizationAlgorithmLevenberg::solve ---- */
        double currentChi = active_robust_chi2(&ba), tempChi = currentChi;
        const double iniChi = currentChi;
        memset(Hpp, 0, sizeof(double) * 36 * (size_t)(nP > 0 ? nP : 1)); memset(Hll, 0, sizeof(double) * 9 * (size_t)(nL > 0 ? nL : 1));
        memset(b, 0, sizeof(double) * (size_t)(sp + sl > 0 ? sp + sl : 1));
        for (int i = 0; i < ba.E; i++) {                                        /* buildSystem: linearize + quadratic form */
            Edge* e = &ba.e[i];
            double A[9], B[18], err[3], rho[3] = { 0, 1, 0 };
            double chi = edge_error(&ba, e, err);
            edge_jacobians(&ba, e, A, B);
            const int D = e->dim;
            double w = e->w;
            if (robust) { huber(chi, D == 2 ? d2 : d3, rho); w *= rho[1]; }       /* weightedOmega = rho[1]*Omega ; omega_r *= rho[1] */
            if (e->point >= 0) {
                double* H = Hll + 9 * (size_t)e->point; double* bb = b + sp + 3 * e->point;
                for (int a = 0; a < 3; a++) {
                    double s = 0; for (int d = 0; d < D; d++) s += A[d * 3 + a] * (-w * err[d]);
                    bb[a] += s;
                    for (int c = 0; c < 3; c++) { double h = 0; for (int d = 0; d < D; d++) h += A[d * 3 + a] * w * A[d * 3 + c]; H[a * 3 + c] += h; }
                }
            }
            if (e->pose >= 0) {
                double* H = Hpp + 36 * (size_t)e->pose; double* bb = b + 6 * e->pose;
                for (int a = 0; a < 6; a++) {
                    double s = 0; for (int d = 0; d < D; d++) s += B[d * 6 + a] * (-w * err[d]);
                    bb[a] += s;
                    for (int c = 0; c < 6; c++) { double h = 0; for (int d = 0; d < D; d++) h += B[d * 6 + a] * w * B[d * 6 + c]; H[a * 6 + c] += h; }
                }
            }
            if (e->pose >= 0 && e->point >= 0) {
                double* H = Hpl + 18 * (size_t)i;
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) { double h = 0; for (int d = 0; d < D; d++) h += B[d * 6 + a] * w * A[d * 3 + c]; H[a * 3 + c] = h; }
            }
        }
        if (it == 0) {                                                          /* computeLambdaInit (:166-180) */
            double maxDiag = 0;
            for (int k = 0; k < nP; k++) for (int j = 0; j < 6; j++) maxDiag = fmax(fabs(Hpp[36 * (size_t)k + 7 * j]), maxDiag);
            for (int l = 0; l < nL; l++) for (int j = 0; j < 3; j++) maxDiag = fmax(fabs(Hll[9 * (size_t)l + 4 * j]), maxDiag);
            lambda = 1e-5 * maxDiag; ni = 2; nBad = 0;
        }
        double rho_lm = 0; int qmax = 0;
        do {
            memcpy(pose_bak, ba.pose, sizeof(SE3) * ba.K); memcpy(pt_bak, ba.pt, sizeof(double) * 3 * ba.M);     /* push() */
            /* setLambda + Schur solve (block_solver.hpp:354-486) */
            int ok2 = 1;
            if (sparse) {
                memset(bs_sys.val, 0, sizeof(double) * 36 * (size_t)bs_sys.rowptr[nP]);
                for (int k = 0; k < nP; k++) { double* Dg = bs_sys.val + 36 * (size_t)bs_sys.rowptr[bs_sys.iperm[k]];
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) Dg[a * 6 + c] = Hpp[36 * (size_t)k + a * 6 + c] + (a == c ? lambda : 0.0); }
            } else {
            for (size_t i = 0; i < (size_t)sp * sp; i++) S[i] = 0;
            for (int k = 0; k < nP; k++) for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                S[(size_t)(6 * k + a) * sp + 6 * k + c] = Hpp[36 * (size_t)k + a * 6 + c] + (a == c ? lambda : 0.0);
            }
            for (int i = 0; i < sp; i++) bs[i] = b[i];
            for (int l = 0; l < nL; l++) {
                double D[9]; memcpy(D, Hll + 9 * (size_t)l, sizeof(D)); D[0] += lambda; D[4] += lambda; D[8] += lambda;
                double* Di = Dinv + 9 * (size_t)l;
                if (!inv3(D, Di)) ok2 = 0;
                double db[3]; const double* bl = b + sp + 3 * l;
                for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * bl[0] + Di[a * 3 + 1] * bl[1] + Di[a * 3 + 2] * bl[2];
                for (int ii = loff[l]; ii < loff[l + 1]; ii++) {
                    const Edge* e1 = &ba.e[ledge[ii]];
                    if (e1->pose < 0) continue;
                    const double* B1 = Hpl + 18 * (size_t)ledge[ii];
                    double BD[18];
                    for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[a * 3 + c] = B1[a * 3] * Di[c] + B1[a * 3 + 1] * Di[3 + c] + B1[a * 3 + 2] * Di[6 + c];
                    for (int a = 0; a < 6; a++) bs[6 * e1->pose + a] -= B1[a * 3] * db[0] + B1[a * 3 + 1] * db[1] + B1[a * 3 + 2] * db[2];
                    for (int jj = loff[l]; jj < loff[l + 1]; jj++) {
                        const Edge* e2 = &ba.e[ledge[jj]];
                        if (e2->pose < 0) continue;
                        const double* B2 = Hpl + 18 * (size_t)ledge[jj];
                        if (sparse) {                                       /* upper triangle of the permuted block matrix (block_solver.hpp:411 keeps j >= i) */
                            const int pi = bs_sys.iperm[e1->pose], qi = bs_sys.iperm[e2->pose];
                            if (pi > qi) continue;
                            double* Sg = bs_sys.val + 36 * (size_t)bsys_slot(&bs_sys, pi, qi);
                            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                                Sg[a * 6 + c] -= BD[a * 3] * B2[c * 3] + BD[a * 3 + 1] * B2[c * 3 + 1] + BD[a * 3 + 2] * B2[c * 3 + 2];
                            continue;
                        }
                        for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                            S[(size_t)(6 * e1->pose + a) * sp + 6 * e2->pose + c] -= BD[a * 3] * B2[c * 3] + BD[a * 3 + 1] * B2[c * 3 + 1] + BD[a * 3 + 2] * B2[c * 3 + 2];
                    }
                }
            }
            for (int i = 0; i < sp; i++) x[i] = bs[i];
            if (sparse && sp > 0 && getenv("ORC_BA_DUMP")) bsys_dump(&bs_sys, x, getenv("ORC_BA_DUMP"));    /* development aid of tools/pcg_proto.py: every reduced system of the call, one file per solve */
            if (sp > 0 && ok2) ok2 = sparse ? bsys_factor_solve(&bs_sys, x) : ldlt_solve(S, sp, x);
            if (ok2) {                                                          /* landmark back-substitution (:456-481) */
                for (int l = 0; l < nL; l++) {
                    double cl[3] = { b[sp + 3 * l], b[sp + 3 * l + 1], b[sp + 3 * l + 2] };
                    for (int ii = loff[l]; ii < loff[l + 1]; ii++) {
                        const Edge* e1 = &ba.e[ledge[ii]];
                        if (e1->pose < 0) continue;
                        const double* B1 = Hpl + 18 * (size_t)ledge[ii]; const double* xp = x + 6 * e1->pose;
                        for (int c = 0; c < 3; c++) { double s = 0; for (int a = 0; a < 6; a++) s += B1[a * 3 + c] * xp[a]; cl[c] -= s; }
                    }
                    const double* Di = Dinv + 9 * (size_t)l;
                    for (int a = 0; a < 3; a++) x[sp + 3 * l + a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
                }
            } else memset(x, 0, sizeof(double) * (size_t)(sp + sl));
            /* update(x): oplus (types_six_dof_expmap.h:73-76, types_sba.h:52-56) */
            for (int k = 0; k < ba.K; k++) if (pidx[k] >= 0) { SE3 ex; se3_exp(x + 6 * pidx[k], &ex); se3_mul(&ex, &ba.pose[k], &ba.pose[k]); }
            for (int m = 0; m < ba.M; m++) if (lidx[m] >= 0) for (int a = 0; a < 3; a++) ba.pt[3 * m + a] += x[sp + 3 * lidx[m] + a];
            tempChi = active_robust_chi2(&ba);
            if (!ok2) tempChi = DBL_MAX;
            rho_lm = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < sp + sl; j++) scale += x[j] * (lambda * x[j] + b[j]);     /* computeScale (:182-189) */
            scale += 1e-3;
            rho_lm /= scale;
            if (rho_lm > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho_lm - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                double sf = fmax(1. / 3., alpha);
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(ba.pose, pose_bak, sizeof(SE3) * ba.K); memcpy(ba.pt, pt_bak, sizeof(double) * 3 * ba.M);  /* pop() */
            }
            qmax++; trials_total++;
        } while (rho_lm < 0 && qmax < 10 && !(stop && *stop));
        it_done++;
        if (r->chi2) r->chi2[it_done] = currentChi;
        if (r->lambda) r->lambda[it_done - 1] = lambda;
        if (qmax == 10 || rho_lm == 0) { ok = 0; continue; }                      /* Terminate */
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;          /* stop criterion (:155-161) */
        if (nBad >= 3) ok = 0;
    }
    *iters_done = it_done; *trials_done = trials_total;
    free(pidx); free(lidx); free(deg); free(ba.e); free(loff); free(ledge);
    if (sparse) bsys_free(&bs_sys);
    free(Hpp); free(Hll); free(Hpl); free(b); free(x); free(S); free(bs); free(Dinv); free(pose_bak); free(pt_bak);
    return 0;
}

static void state_from_floats(const OrcBAProblem* p, SE3* pose, double* pt)
{
    for (int k = 0; k < p->n_poses; k++) {                                       /* Converter::toSE3Quat (Converter.cc:37-47) */
        const float* T = p->poses + 16 * k;
        double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
        quat_from_R(R, pose[k].q); quat_normalize(pose[k].q);
        pose[k].t[0] = T[3]; pose[k].t[1] = T[7]; pose[k].t[2] = T[11];
    }
    for (int m = 0; m < 3 * p->n_points; m++) pt[m] = p->points[m];
}

/* write back (Converter::toCvMat: double -> float); vertices that were never optimised are passed through */
static void state_to_floats(const OrcBAProblem* p, const SE3* pose, const double* pt, const uint8_t* pose_touched, const uint8_t* pt_touched, OrcBAResult* r)
{
    for (int k = 0; k < p->n_poses; k++) {
        float* T = r->poses + 16 * k;
        if (p->pose_fixed[k] || (pose_touched && !pose_touched[k])) { memcpy(T, p->poses + 16 * k, sizeof(float) * 16); continue; }
        double R[9]; quat_to_R(pose[k].q, R);
        T[0] = (float)R[0]; T[1] = (float)R[1]; T[2] = (float)R[2]; T[3] = (float)pose[k].t[0];
        T[4] = (float)R[3]; T[5] = (float)R[4]; T[6] = (float)R[5]; T[7] = (float)pose[k].t[1];
        T[8] = (float)R[6]; T[9] = (float)R[7]; T[10] = (float)R[8]; T[11] = (float)pose[k].t[2];
        T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    }
    for (int m = 0; m < p->n_points; m++) {
        const int keep = p->point_fixed[m] || (pt_touched && !pt_touched[m]);
        for (int a = 0; a < 3; a++) r->points[3 * m + a] = keep ? p->points[3 * m + a] : (float)pt[3 * m + a];
    }
}

int orc_ba_solve(const OrcBAProblem* p, int iters, int robust, volatile int* stop, OrcBAResult* r)
{
    SE3* pose = (SE3*)malloc(sizeof(SE3) * (p->n_poses > 0 ? p->n_poses : 1));
    double* pt = (double*)malloc(sizeof(double) * 3 * (p->n_points > 0 ? p->n_points : 1));
    state_from_floats(p, pose, pt);
    uint8_t* ptt = (uint8_t*)calloc(p->n_points > 0 ? p->n_points : 1, 1);          /* points without edges are removed (Optimizer.cc:198-202) */
    for (int i = 0; i < p->n_edges; i++) if (!(p->pose_fixed[p->edges[i].pose] && p->point_fixed[p->edges[i].point])) ptt[p->edges[i].point] = 1;
    /* thHuber2D = sqrt(5.99), thHuber3D = sqrt(7.815) as floats (Optimizer.cc:102-103) */
    double* cam = cam_table(p);
    int rc = ba_optimize(p, cam, NULL, pose, pt, iters, robust, stop, r->chi2, r->lambda, &r->iters_done, &r->trials_total, NULL,
                         (double)(float)sqrt(5.99), (double)(float)sqrt(7.815));
    if (rc == 0) state_to_floats(p, pose, pt, NULL, ptt, r);
    free(pose); free(pt); free(ptt); free(cam);
    return rc;
}

/* Multi-stage optimisation with per-edge outlier classification between stages.
 * Optimizer::LocalBundleAdjustment (Optimizer.cc:487-838): stages {5 it, robust}, {10 it, non-robust}; after each stage an
 *   edge becomes inactive (setLevel(1)) if chi2 > 5.991 / 7.815 or depth <= 0; chi2 is the STALE value of the edge's last
 *   computeError() (g2o does not refresh _error after optimize()), depth is evaluated fresh (isDepthPositive()).
 * Optimizer::PoseOptimization (Optimizer.cc:272-485): 4 stages of 10 iterations, robust except the last, the estimate is
 *   reset to the input before every stage, inactive edges get a FRESH error before the test and may become active again,
 *   the comparison is done in float.  Flags select these behaviours. */
int orc_ba_solve_staged(const OrcBAProblem* p, const OrcBAStage* st, int n_stages, volatile int* stop, OrcBAResult* r, uint8_t* edge_outlier)
{
    SE3* pose = (SE3*)malloc(sizeof(SE3) * (p->n_poses > 0 ? p->n_poses : 1));
    double* pt = (double*)malloc(sizeof(double) * 3 * (p->n_points > 0 ? p->n_points : 1));
    SE3* pose0 = (SE3*)malloc(sizeof(SE3) * (p->n_poses > 0 ? p->n_poses : 1));
    double* pt0 = (double*)malloc(sizeof(double) * 3 * (p->n_points > 0 ? p->n_points : 1));
    state_from_floats(p, pose, pt);
    memcpy(pose0, pose, sizeof(SE3) * p->n_poses); memcpy(pt0, pt, sizeof(double) * 3 * p->n_points);
    const int E = p->n_edges;
    uint8_t* active = (uint8_t*)malloc(E > 0 ? E : 1); memset(active, 1, E > 0 ? E : 1);
    double* last = (double*)calloc(E > 0 ? E : 1, sizeof(double));
    uint8_t* pose_t = (uint8_t*)calloc(p->n_poses > 0 ? p->n_poses : 1, 1), * pt_t = (uint8_t*)calloc(p->n_points > 0 ? p->n_points : 1, 1);
    int rc = 0, its = 0, trials = 0;
    r->iters_done = 0; r->trials_total = 0;
    double* cam = cam_table(p);
    BA ev; memset(&ev, 0, sizeof(ev)); ev.cam = cam; ev.pose = pose; ev.pt = pt;
    if (stop && *stop) {                                                          /* `if(pbStopFlag) if(*pbStopFlag) return;` (Optimizer.cc:706-708): nothing is touched */
        memcpy(r->poses, p->poses, sizeof(float) * 16 * (size_t)p->n_poses); memcpy(r->points, p->points, sizeof(float) * 3 * (size_t)p->n_points);
        if (edge_outlier) memset(edge_outlier, 0, E > 0 ? E : 0);
        free(pose); free(pt); free(pose0); free(pt0); free(active); free(last); free(pose_t); free(pt_t); free(cam);
        return 0;
    }
    for (int s = 0; s < n_stages && rc == 0; s++) {
        if (st[s].reset_estimates) { memcpy(pose, pose0, sizeof(SE3) * p->n_poses); memcpy(pt, pt0, sizeof(double) * 3 * p->n_points); }
        for (int i = 0; i < E; i++) if (active[i] && !(p->pose_fixed[p->edges[i].pose] && p->point_fixed[p->edges[i].point])) { pose_t[p->edges[i].pose] = 1; pt_t[p->edges[i].point] = 1; }
        rc = ba_optimize(p, cam, active, pose, pt, st[s].iterations, st[s].robust, stop, NULL, NULL, &its, &trials, last,
                         (double)st[s].huber_mono, (double)st[s].huber_stereo);
        r->iters_done += its; r->trials_total += trials;
        /* stop flag raised during / after this optimize(): bDoMore = false skips the remaining rounds, the final "Check inlier observations"
         * pass (the LAST stage's test, every edge, stale chi2, fresh depth) and the write-back still run (Optimizer.cc:712-800) */
        const int stopped = stop && *stop;
        const OrcBAStage* cs = stopped ? &st[n_stages - 1] : &st[s];
        for (int i = 0; i < E; i++) {                                             /* classification */
            const OrcBAEdge* e = &p->edges[i];
            Edge ed; ed.vpose = e->pose; ed.vpoint = e->point; ed.dim = e->ur < 0 ? 2 : 3; ed.obs[0] = e->u; ed.obs[1] = e->v; ed.obs[2] = e->ur; ed.w = e->inv_sigma2;
            double err[3];
            if (!active[i] && cs->recompute_inactive) last[i] = edge_error(&ev, &ed, err);
            if (!active[i] && !cs->allow_reactivate) continue;
            const float thf = ed.dim == 2 ? cs->chi2_mono : cs->chi2_stereo;
            const double thd = round((double)thf * 1e6) / 1e6;                    /* the decimal literal 5.991 / 7.815 as the double the reference compares with */
            int out = cs->float_compare ? ((float)last[i] > thf) : (last[i] > thd);
            if (cs->check_depth) {
                double Xc[3]; quat_rot(pose[e->pose].q, pt + 3 * e->point, Xc);
                if (!(Xc[2] + pose[e->pose].t[2] > 0.0)) out = 1;
            }
            active[i] = out ? 0 : 1;
        }
        if (stopped) break;
    }
    if (edge_outlier) for (int i = 0; i < E; i++) edge_outlier[i] = active[i] ? 0 : 1;
    if (rc == 0) state_to_floats(p, pose, pt, pose_t, pt_t, r);
    free(pose); free(pt); free(pose0); free(pt0); free(active); free(last); free(pose_t); free(pt_t); free(cam);
    return rc;
}
