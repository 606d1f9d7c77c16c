/* CPU ORACLE (test infrastructure) -- Optimizer::OptimizeSim3 (corbslam_client/src/Optimizer.cc:1119-1311) and
 * Optimizer::OptimizeEssentialGraph (Optimizer.cc:840-1117; EdgeSim3, BlockSolver_7_3, lambda_init 1e-16, 20 iterations).
 * Restates the g2o pieces it drives (G/ = corbslam_client/Thirdparty/g2o/g2o/):
 *   G/types/sim3.h:40-250                      Sim3: exp-map constructor, operator*, inverse, map
 *   G/types/types_seven_dof_expmap.h:50-170    VertexSim3Expmap::oplusImpl (_fix_scale), EdgeSim3ProjectXYZ, EdgeInverseSim3ProjectXYZ
 *   G/core/base_binary_edge.hpp:131-200        numeric Jacobian (delta = 1e-9, central differences) -- the two edges define no linearizeOplus
 *   G/core/optimization_algorithm_levenberg.cpp:61-189, G/core/robust_kernel_impl.cpp:78-91, G/solvers/linear_solver_dense.h:65-112
 * One free 7-dof vertex, fixed points: the system is a single 7x7 block solved by a dense LDL^T (Eigen::LDLT there, no
 * pivoting here -- same factorisation of a positive definite matrix up to rounding).  Tolerance 1e-4 relative, see orc.h. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct { double q[4]; double t[3]; double s; } Sim3;     /* q = x y z w, NOT re-normalised (sim3.h never does) */

static void q_from_R(const double R[9], double* q)              /* Eigen::Quaterniond(Matrix3d) */
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) { t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
static void q_to_R(const double* q, double R[9])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void q_rot(const double* q, const double* v, double* o)
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void q_mul(const double* a, const double* b, double* o)
{
    double r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    memcpy(o, r, sizeof(r));
}

/* Sim3(const Vector7d& update) (sim3.h:73-150): update = (omega, upsilon, sigma) */
static void sim3_exp(const double* u, Sim3* S)
{
    const double om[3] = { u[0], u[1], u[2] }, up[3] = { u[3], u[4], u[5] }, sigma = u[6];
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += O[i * 3 + k] * O[k * 3 + j]; O2[i * 3 + j] = s; }
    const double s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C, R[9];
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; for (int i = 0; i < 9; i++) R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + O[i] + O2[i]; }
        else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta);
            for (int i = 0; i < 9; i++) R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + sin(theta) / theta * O[i] + (1 - cos(theta)) / (theta * theta) * O2[i];
        }
    } else {
        C = (s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
            for (int i = 0; i < 9; i++) R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + O[i] + O2[i];
        } else {
            for (int i = 0; i < 9; i++) R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + sin(theta) / theta * O[i] + (1 - cos(theta)) / (theta * theta) * O2[i];
            const double a = s * sin(theta), b = s * cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    q_from_R(R, S->q);
    for (int i = 0; i < 3; i++) {
        double acc = 0;
        for (int j = 0; j < 3; j++) acc += (A * O[i * 3 + j] + B * O2[i * 3 + j] + C * (i == j ? 1.0 : 0.0)) * up[j];
        S->t[i] = acc;
    }
    S->s = s;
}
static void sim3_mul(const Sim3* a, const Sim3* b, Sim3* o)
{
    Sim3 r; double rt[3];
    q_mul(a->q, b->q, r.q);
    q_rot(a->q, b->t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a->s * rt[i] + a->t[i];
    r.s = a->s * b->s;
    *o = r;
}
static void sim3_inv(const Sim3* a, Sim3* o)
{
    Sim3 r; r.q[0] = -a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = a->q[3];
    const double k = -1. / a->s; const double v[3] = { k * a->t[0], k * a->t[1], k * a->t[2] };
    q_rot(r.q, v, r.t);
    r.s = 1. / a->s;
    *o = r;
}
static void sim3_map(const Sim3* S, const double* x, double* o)
{
    double r[3]; q_rot(S->q, x, r);
    for (int i = 0; i < 3; i++) o[i] = S->s * r[i] + S->t[i];
}

typedef struct {
    const OrcSim3Problem* p;
    int n; const uint8_t* alive;          /* pairs still in the graph */
    double delta;                         /* Huber */
    double* last12; double* last21;       /* chi2 of the last computeError() per edge */
} S3;

static void pair_errors(const S3* g, int i, const Sim3* S, const Sim3* Sinv, double* e12, double* e21)
{
    const OrcSim3Problem* p = g->p;
    const double X1[3] = { p->p1c[3 * i], p->p1c[3 * i + 1], p->p1c[3 * i + 2] }, X2[3] = { p->p2c[3 * i], p->p2c[3 * i + 1], p->p2c[3 * i + 2] };
    double m[3];
    sim3_map(S, X2, m);                                              /* x1 = S12 * X2 */
    e12[0] = (double)p->obs1[2 * i] - (m[0] / m[2] * (double)p->fx1 + (double)p->cx1);
    e12[1] = (double)p->obs1[2 * i + 1] - (m[1] / m[2] * (double)p->fy1 + (double)p->cy1);
    sim3_map(Sinv, X1, m);                                           /* x2 = S12^-1 * X1 */
    e21[0] = (double)p->obs2[2 * i] - (m[0] / m[2] * (double)p->fx2 + (double)p->cx2);
    e21[1] = (double)p->obs2[2 * i + 1] - (m[1] / m[2] * (double)p->fy2 + (double)p->cy2);
}
static void huber7(double e, double delta, double* rho)
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; }
}
static double robust_chi2(const S3* g, const Sim3* S)
{
    Sim3 Si; sim3_inv(S, &Si);
    double chi = 0;
    for (int i = 0; i < g->n; i++) {
        if (!g->alive[i]) continue;
        double e12[2], e21[2], rho[2];
        pair_errors(g, i, S, &Si, e12, e21);
        const double c12 = (double)g->p->inv_sigma2_1[i] * (e12[0] * e12[0] + e12[1] * e12[1]);
        const double c21 = (double)g->p->inv_sigma2_2[i] * (e21[0] * e21[0] + e21[1] * e21[1]);
        g->last12[i] = c12; g->last21[i] = c21;
        huber7(c12, g->delta, rho); chi += rho[0];
        huber7(c21, g->delta, rho); chi += rho[0];
    }
    return chi;
}
static int ldlt7(double* a, double* b)
{
    const int n = 7;
    for (int j = 0; j < n; j++) {
        double d = a[j * n + j];
        for (int k = 0; k < j; k++) d -= a[j * n + k] * a[j * n + k] * a[k * n + k];
        if (!isfinite(d) || d <= 0.0) return 0;                      /* Eigen::LDLT::isPositive() */
        a[j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double s = a[i * n + j]; for (int k = 0; k < j; k++) s -= a[i * n + k] * a[j * n + k] * a[k * n + k]; a[i * n + j] = s / d; }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= a[i * n + k] * b[k]; b[i] = s; }
    for (int i = 0; i < n; i++) b[i] /= a[i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= a[k * n + i] * b[k]; b[i] = s; }
    return 1;
}
static void oplus(Sim3* S, const double* x, int fix_scale)
{
    double u[7]; memcpy(u, x, sizeof(u));
    if (fix_scale) u[6] = 0;
    Sim3 e; sim3_exp(u, &e);
    sim3_mul(&e, S, S);
}

/* optimizer.initializeOptimization(); optimizer.optimize(iters) on the alive pairs */
static void optimize7(S3* g, Sim3* S, int iters, int fix_scale, int* iters_done, int* trials)
{
    double lambda = -1, ni = 2; int nBad = 0, ok = 1;
    for (int it = 0; it < iters && ok; it++) {
        double currentChi = robust_chi2(g, S), tempChi;
        const double iniChi = currentChi;
        double H[49], b[7]; memset(H, 0, sizeof(H)); memset(b, 0, sizeof(b));
        /* the 14 perturbed states of the numeric Jacobian are the same for every edge */
        Sim3 Sp[7], Sm[7], Spi[7], Smi[7], Si; sim3_inv(S, &Si);
        for (int d = 0; d < 7; d++) {
            double u[7] = { 0, 0, 0, 0, 0, 0, 0 };
            u[d] = 1e-9; Sp[d] = *S; oplus(&Sp[d], u, fix_scale); sim3_inv(&Sp[d], &Spi[d]);
            u[d] = -1e-9; Sm[d] = *S; oplus(&Sm[d], u, fix_scale); sim3_inv(&Sm[d], &Smi[d]);
        }
        const double scalar = 1.0 / (2 * 1e-9);
        for (int i = 0; i < g->n; i++) {
            if (!g->alive[i]) continue;
            double e12[2], e21[2], J12[14], J21[14];
            pair_errors(g, i, S, &Si, e12, e21);
            for (int d = 0; d < 7; d++) {
                double a12[2], a21[2], b12[2], b21[2];
                pair_errors(g, i, &Sp[d], &Spi[d], a12, a21);
                pair_errors(g, i, &Sm[d], &Smi[d], b12, b21);
                J12[d] = scalar * (a12[0] - b12[0]); J12[7 + d] = scalar * (a12[1] - b12[1]);
                J21[d] = scalar * (a21[0] - b21[0]); J21[7 + d] = scalar * (a21[1] - b21[1]);
            }
            for (int k = 0; k < 2; k++) {
                const double* e = k ? e21 : e12; const double* J = k ? J21 : J12;
                double w = k ? (double)g->p->inv_sigma2_2[i] : (double)g->p->inv_sigma2_1[i], rho[2];
                huber7(w * (e[0] * e[0] + e[1] * e[1]), g->delta, rho);
                w *= rho[1];
                for (int a = 0; a < 7; a++) {
                    b[a] += J[a] * (-w * e[0]) + J[7 + a] * (-w * e[1]);
                    for (int c = 0; c < 7; c++) H[a * 7 + c] += J[a] * w * J[c] + J[7 + a] * w * J[7 + c];
                }
            }
        }
        if (it == 0) { double maxDiag = 0; for (int j = 0; j < 7; j++) maxDiag = fmax(fabs(H[j * 8]), maxDiag); lambda = 1e-5 * maxDiag; ni = 2; nBad = 0; }
        double rho_lm = 0; int qmax = 0;
        do {
            const Sim3 bak = *S;
            double A[49], x[7]; memcpy(A, H, sizeof(A)); memcpy(x, b, sizeof(x));
            for (int j = 0; j < 7; j++) A[j * 8] += lambda;
            int ok2 = ldlt7(A, x);
            if (!ok2) memset(x, 0, sizeof(x));
            oplus(S, x, fix_scale);
            tempChi = robust_chi2(g, S);
            if (!ok2) tempChi = DBL_MAX;
            rho_lm = currentChi - tempChi;
            double scale = 0; for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3; rho_lm /= scale;
            if (rho_lm > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho_lm - 1), 3); alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; *S = bak; }
            qmax++; (*trials)++;
        } while (rho_lm < 0 && qmax < 10);
        (*iters_done)++;
        if (qmax == 10 || rho_lm == 0) { ok = 0; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) ok = 0;
    }
}

int orc_optimize_sim3(const OrcSim3Problem* p, double* R12, double* t12, double* s12, float th2, int fix_scale, uint8_t* removed,
                      int* iters_done, int* trials)
{
    const int n = p->n;
    Sim3 S; q_from_R(R12, S.q); S.t[0] = t12[0]; S.t[1] = t12[1]; S.t[2] = t12[2]; S.s = *s12;   /* Sim3(Matrix3d R, t, s) */
    uint8_t* alive = (uint8_t*)malloc(n > 0 ? n : 1);
    double* l12 = (double*)calloc(n > 0 ? n : 1, sizeof(double)); double* l21 = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    for (int i = 0; i < n; i++) { alive[i] = 1; removed[i] = 0; }
    S3 g; g.p = p; g.n = n; g.alive = alive; g.delta = (double)sqrtf(th2); g.last12 = l12; g.last21 = l21;     /* const float deltaHuber = sqrt(th2) */
    int it = 0, tr = 0;
    optimize7(&g, &S, 5, fix_scale, &it, &tr);
    int nBad = 0;
    for (int i = 0; i < n; i++) if (l12[i] > (double)th2 || l21[i] > (double)th2) { removed[i] = 1; alive[i] = 0; nBad++; }
    const int nMore = nBad > 0 ? 10 : 5;
    int nIn = 0;
    if (n - nBad >= 10) {
        optimize7(&g, &S, nMore, fix_scale, &it, &tr);
        for (int i = 0; i < n; i++) {
            if (!alive[i]) continue;
            if (l12[i] > (double)th2 || l21[i] > (double)th2) removed[i] = 1; else nIn++;
        }
        q_to_R(S.q, R12); t12[0] = S.t[0]; t12[1] = S.t[1]; t12[2] = S.t[2]; *s12 = S.s;          /* g2oS12 = vSim3_recov->estimate() */
    }
    if (iters_done) *iters_done = it;
    if (trials) *trials = tr;
    free(alive); free(l12); free(l21);
    return nIn;
}

/* =====================================================================================================================
 * Optimizer::OptimizeEssentialGraph (Optimizer.cc:840-1117): pose graph over Sim3 vertices, EdgeSim3 with identity
 * information, no robust kernel, numeric Jacobians for BOTH vertices, Levenberg with setUserLambdaInit(1e-16), 20 iterations.
 * The adapter builds the graph (which keyframes, which edges, measurements Sji = Sjw * Swi); here: the optimisation, the
 * SE3 recovery [R t/s] and the map point correction through the reference keyframe.
 * The reduced system (free vertices x 7) is solved by a dense LDL^T (reference: Eigen SimplicialLDLT via LinearSolverEigen). */

/* Sim3::log() (sim3.h:158-232) */
static void solve3_lu(const double* W, const double* b, double* x)       /* W.lu().solve(t): partial pivoting */
{
    double a[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) a[i][j] = W[i * 3 + j]; a[i][3] = b[i]; }
    for (int c = 0; c < 3; c++) {
        int piv = c; for (int r = c + 1; r < 3; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 4; j++) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        for (int r = c + 1; r < 3; r++) { const double f = a[r][c] / a[c][c]; for (int j = c; j < 4; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 2; i >= 0; i--) { double s = a[i][3]; for (int j = i + 1; j < 3; j++) s -= a[i][j] * x[j]; x[i] = s / a[i][i]; }
}
static void sim3_log(const Sim3* S, double* res)
{
    const double s = S->s, sigma = log(s);
    double R[9]; q_to_R(S->q, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };       /* deltaR */
    const double eps = 0.00001;
    double A, B, C, om[3];
    if (fabs(sigma) < eps) {
        C = 1;
        if (d > 1 - eps) { for (int i = 0; i < 3; i++) om[i] = 0.5 * dR[i]; A = 1. / 2.; B = 1. / 6.; }
        else {
            const double theta = acos(d), theta2 = theta * theta, k = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; i++) om[i] = k * dR[i];
            A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta);
        }
    } else {
        C = (s - 1) / sigma;
        if (d > 1 - eps) {
            const double sigma2 = sigma * sigma;
            for (int i = 0; i < 3; i++) om[i] = 0.5 * dR[i];
            A = ((sigma - 1) * s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * s) / (sigma2 * sigma);
        } else {
            const double theta = acos(d), k = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; i++) om[i] = k * dR[i];
            const double theta2 = theta * theta, a = s * sin(theta), b = s * cos(theta), c = theta2 + sigma * sigma;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9], W[9], up[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += O[i * 3 + k] * O[k * 3 + j]; O2[i * 3 + j] = t; }
    for (int i = 0; i < 9; i++) W[i] = A * O[i] + B * O2[i] + C * ((i % 4) == 0 ? 1.0 : 0.0);
    solve3_lu(W, S->t, up);
    res[0] = om[0]; res[1] = om[1]; res[2] = om[2]; res[3] = up[0]; res[4] = up[1]; res[5] = up[2]; res[6] = sigma;
}
/* EdgeSim3::computeError: error = log(C * v1 * v2^-1), v1 = vertex(0) = i, v2 = vertex(1) = j */
static void edge_sim3_error(const Sim3* C, const Sim3* Si, const Sim3* Sj, double* e)
{
    Sim3 Sjinv, t1, t2; sim3_inv(Sj, &Sjinv); sim3_mul(C, Si, &t1); sim3_mul(&t1, &Sjinv, &t2); sim3_log(&t2, e);
}
static void load8(const double* v, Sim3* S) { S->q[0] = v[0]; S->q[1] = v[1]; S->q[2] = v[2]; S->q[3] = v[3]; S->t[0] = v[4]; S->t[1] = v[5]; S->t[2] = v[6]; S->s = v[7]; }
static void store8(const Sim3* S, double* v) { v[0] = S->q[0]; v[1] = S->q[1]; v[2] = S->q[2]; v[3] = S->q[3]; v[4] = S->t[0]; v[5] = S->t[1]; v[6] = S->t[2]; v[7] = S->s; }

static int ldlt_dense(double* a, int n, double* b)
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

/* vertices: K x 8 doubles (quaternion x y z w, t, s) in / out; fixed[K]; edges (vi[e], vj[e]) = (vertex 0, vertex 1), meas E x 8.
 * chi2_hist (iters + 1, may be NULL).  Returns 0, or -2 for an edge index out of range. */
int orc_optimize_essential_graph(int K, double* S, const uint8_t* fixed, int E, const int32_t* vi, const int32_t* vj, const double* meas,
                                 int iters, int fix_scale, double* chi2_hist, int* iters_done, int* trials_done)
{
    for (int e = 0; e < E; e++) if (vi[e] < 0 || vi[e] >= K || vj[e] < 0 || vj[e] >= K) return -2;
    int* idx = (int*)malloc(sizeof(int) * (K > 0 ? K : 1)); int nP = 0;
    for (int k = 0; k < K; k++) idx[k] = fixed[k] ? -1 : nP++;
    const int sp = 7 * nP;
    Sim3* V = (Sim3*)malloc(sizeof(Sim3) * (K > 0 ? K : 1)); Sim3* bak = (Sim3*)malloc(sizeof(Sim3) * (K > 0 ? K : 1));
    for (int k = 0; k < K; k++) load8(S + 8 * (size_t)k, &V[k]);
    double* H = (double*)malloc(sizeof(double) * (size_t)(sp > 0 ? sp : 1) * (sp > 0 ? sp : 1));
    double* A = (double*)malloc(sizeof(double) * (size_t)(sp > 0 ? sp : 1) * (sp > 0 ? sp : 1));
    double* b = (double*)malloc(sizeof(double) * (sp > 0 ? sp : 1)); double* x = (double*)malloc(sizeof(double) * (sp > 0 ? sp : 1));
    double lambda = 1e-16, ni = 2; int nBad = 0, ok = 1, it_done = 0, trials = 0;
    #define CHI2(out) do { double c_ = 0; for (int e_ = 0; e_ < E; e_++) { if (fixed[vi[e_]] && fixed[vj[e_]]) continue; Sim3 C_; load8(meas + 8 * (size_t)e_, &C_); double er_[7]; \
        edge_sim3_error(&C_, &V[vi[e_]], &V[vj[e_]], er_); for (int q_ = 0; q_ < 7; q_++) c_ += er_[q_] * er_[q_]; } (out) = c_; } while (0)
    if (chi2_hist) CHI2(chi2_hist[0]);
    for (int it = 0; it < iters && ok && nP > 0; it++) {
        double currentChi, tempChi; CHI2(currentChi);
        const double iniChi = currentChi;
        memset(H, 0, sizeof(double) * (size_t)sp * sp); memset(b, 0, sizeof(double) * sp);
        const double scalar = 1.0 / (2 * 1e-9);
        for (int e = 0; e < E; e++) {
            const int a = vi[e], c = vj[e];
            if (fixed[a] && fixed[c]) continue;
            Sim3 C; load8(meas + 8 * (size_t)e, &C);
            double err[7], Ji[49], Jj[49];
            edge_sim3_error(&C, &V[a], &V[c], err);
            for (int side = 0; side < 2; side++) {
                const int v = side ? c : a; double* J = side ? Jj : Ji;
                if (fixed[v]) continue;
                for (int d = 0; d < 7; d++) {
                    double u[7] = { 0, 0, 0, 0, 0, 0, 0 }, ep[7], em[7];
                    Sim3 P = V[v]; u[d] = 1e-9; oplus(&P, u, fix_scale);
                    edge_sim3_error(&C, side ? &V[a] : &P, side ? &P : &V[c], ep);
                    P = V[v]; u[d] = -1e-9; oplus(&P, u, fix_scale);
                    edge_sim3_error(&C, side ? &V[a] : &P, side ? &P : &V[c], em);
                    for (int r = 0; r < 7; r++) J[r * 7 + d] = scalar * (ep[r] - em[r]);
                }
            }
            const int ia = idx[a], ic = idx[c];
            if (ia >= 0) for (int p = 0; p < 7; p++) {
                double s = 0; for (int r = 0; r < 7; r++) s += Ji[r * 7 + p] * (-err[r]);
                b[7 * ia + p] += s;
                for (int q = 0; q < 7; q++) { double h = 0; for (int r = 0; r < 7; r++) h += Ji[r * 7 + p] * Ji[r * 7 + q]; H[(size_t)(7 * ia + p) * sp + 7 * ia + q] += h; }
            }
            if (ic >= 0) for (int p = 0; p < 7; p++) {
                double s = 0; for (int r = 0; r < 7; r++) s += Jj[r * 7 + p] * (-err[r]);
                b[7 * ic + p] += s;
                for (int q = 0; q < 7; q++) { double h = 0; for (int r = 0; r < 7; r++) h += Jj[r * 7 + p] * Jj[r * 7 + q]; H[(size_t)(7 * ic + p) * sp + 7 * ic + q] += h; }
            }
            if (ia >= 0 && ic >= 0) for (int p = 0; p < 7; p++) for (int q = 0; q < 7; q++) {
                double h = 0; for (int r = 0; r < 7; r++) h += Ji[r * 7 + p] * Jj[r * 7 + q];
                H[(size_t)(7 * ia + p) * sp + 7 * ic + q] += h; H[(size_t)(7 * ic + q) * sp + 7 * ia + p] += h;
            }
        }
        if (it == 0) { lambda = 1e-16; ni = 2; nBad = 0; }                     /* setUserLambdaInit(1e-16) */
        double rho_lm = 0; int qmax = 0;
        do {
            memcpy(bak, V, sizeof(Sim3) * K);
            memcpy(A, H, sizeof(double) * (size_t)sp * sp);
            for (int j = 0; j < sp; j++) { A[(size_t)j * sp + j] += lambda; x[j] = b[j]; }
            int ok2 = ldlt_dense(A, sp, x);
            if (!ok2) memset(x, 0, sizeof(double) * sp);
            for (int k = 0; k < K; k++) if (idx[k] >= 0) oplus(&V[k], x + 7 * idx[k], fix_scale);
            CHI2(tempChi);
            if (!ok2) tempChi = DBL_MAX;
            rho_lm = currentChi - tempChi;
            double scale = 0; for (int j = 0; j < sp; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3; rho_lm /= scale;
            if (rho_lm > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho_lm - 1), 3); alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; memcpy(V, bak, sizeof(Sim3) * K); }
            qmax++; trials++;
        } while (rho_lm < 0 && qmax < 10);
        it_done++;
        if (chi2_hist) chi2_hist[it_done] = currentChi;
        if (qmax == 10 || rho_lm == 0) { ok = 0; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) ok = 0;
    }
    #undef CHI2
    for (int k = 0; k < K; k++) store8(&V[k], S + 8 * (size_t)k);
    if (iters_done) *iters_done = it_done;
    if (trials_done) *trials_done = trials;
    free(idx); free(V); free(bak); free(H); free(A); free(b); free(x);
    return 0;
}

/* after the optimisation (Optimizer.cc:1045-1114): SE3 recovery Tiw = [R | t/s] (Converter::toCvSE3: double -> float) and map point
 * correction p <- correctedSwr.map(Srw.map(p)) with the reference keyframe's old / corrected similarity (ref < 0: untouched) */
void orc_essential_graph_apply(int K, const double* S_old, const double* S_new, float* Tiw_out, int M, const int32_t* ref, float* points)
{
    for (int k = 0; k < K; k++) {
        Sim3 S; load8(S_new + 8 * (size_t)k, &S);
        double R[9]; q_to_R(S.q, R);
        const double is = 1. / S.s;
        float* T = Tiw_out + 16 * (size_t)k;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i * 3 + j]; T[i * 4 + 3] = (float)(S.t[i] * is); }
        T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    }
    for (int m = 0; m < M; m++) {
        if (ref[m] < 0 || ref[m] >= K) continue;
        Sim3 Srw, Snew, Swr; load8(S_old + 8 * (size_t)ref[m], &Srw); load8(S_new + 8 * (size_t)ref[m], &Snew); sim3_inv(&Snew, &Swr);
        const double p[3] = { points[3 * m], points[3 * m + 1], points[3 * m + 2] };
        double a[3], c[3]; sim3_map(&Srw, p, a); sim3_map(&Swr, a, c);
        points[3 * m] = (float)c[0]; points[3 * m + 1] = (float)c[1]; points[3 * m + 2] = (float)c[2];
    }
}
