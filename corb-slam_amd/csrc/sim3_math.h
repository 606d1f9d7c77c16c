// sim3_math.h -- g2o::Sim3 on the device (G/types/sim3.h): exp-map constructor, product, inverse, log; quaternion x y z w, never
// re-normalised (like the reference).  Shared by OptimizeSim3 and OptimizeEssentialGraph.  Device code only.
#pragma once
#include "ba_math.h"

struct S3State { double q[4], t[3], s; };

__device__ __forceinline__ void s3_qmul(const double* a, const double* b, double* o)
{
    double r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
// Sim3(const Vector7d& update) (G/types/sim3.h:73-150)
__device__ void s3_exp(const double* u, S3State& S)
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
    quat_from_R(R, S.q);
    for (int i = 0; i < 3; i++) {
        double acc = 0;
        for (int j = 0; j < 3; j++) acc += (A * O[i * 3 + j] + B * O2[i * 3 + j] + C * (i == j ? 1.0 : 0.0)) * up[j];
        S.t[i] = acc;
    }
    S.s = s;
}
__device__ __forceinline__ void s3_mul(const S3State& a, const S3State& b, S3State& o)
{
    S3State r; double rt[3];
    s3_qmul(a.q, b.q, r.q);
    quat_rot(a.q, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.s * rt[i] + a.t[i];
    r.s = a.s * b.s;
    o = r;
}
__device__ __forceinline__ void s3_inv(const S3State& a, S3State& o)
{
    S3State r; r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
    const double k = -1. / a.s; const double v[3] = { k * a.t[0], k * a.t[1], k * a.t[2] };
    quat_rot(r.q, v, r.t);
    r.s = 1. / a.s;
    o = r;
}
__device__ __forceinline__ void s3_oplus(S3State& S, const double* x, int fix_scale)
{
    double u[7]; for (int i = 0; i < 7; i++) u[i] = x[i];
    if (fix_scale) u[6] = 0;
    S3State e; s3_exp(u, e);
    s3_mul(e, S, S);
}
// Sim3::map (sim3.h:152-154)
__device__ __forceinline__ void s3_map(const S3State& S, const double* x, double* o)
{
    double r[3]; quat_rot(S.q, x, r);
    for (int i = 0; i < 3; i++) o[i] = S.s * r[i] + S.t[i];
}
// W.lu().solve(t) of a 3x3 system: Gaussian elimination with partial pivoting (Eigen::PartialPivLU)
__device__ __forceinline__ void s3_solve3_lu(const double* W, const double* b, double* x)
{
    double a[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) a[i][j] = W[i * 3 + j]; a[i][3] = b[i]; }
    for (int c = 0; c < 3; c++) {
        int piv = c; for (int r = c + 1; r < 3; r++) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 4; j++) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        for (int r = c + 1; r < 3; r++) { const double f = a[r][c] / a[c][c]; for (int j = c; j < 4; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 2; i >= 0; i--) { double s = a[i][3]; for (int j = i + 1; j < 3; j++) s -= a[i][j] * x[j]; x[i] = s / a[i][i]; }
}
// Sim3::log() (sim3.h:158-232)
__device__ void s3_log(const S3State& S, double* res)
{
    const double s = S.s, sigma = log(s);
    double R[9]; quat_to_R(S.q, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };
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
    s3_solve3_lu(W, S.t, up);
    res[0] = om[0]; res[1] = om[1]; res[2] = om[2]; res[3] = up[0]; res[4] = up[1]; res[5] = up[2]; res[6] = sigma;
}
__device__ __forceinline__ void s3_load(const double* v, S3State& S) { for (int k = 0; k < 4; k++) S.q[k] = v[k]; for (int k = 0; k < 3; k++) S.t[k] = v[4 + k]; S.s = v[7]; }
__device__ __forceinline__ void s3_store(const S3State& S, double* v) { for (int k = 0; k < 4; k++) v[k] = S.q[k]; for (int k = 0; k < 3; k++) v[4 + k] = S.t[k]; v[7] = S.s; }
