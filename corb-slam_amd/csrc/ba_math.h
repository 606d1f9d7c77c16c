// ba_math.h -- SE3 / quaternion helpers shared by the bundle-adjustment kernels (Eigen 3 formulas, see oracle/orc_ba.c;
// G/types/se3quat.h, se3_ops.hpp).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void quat_to_R(const double* q, double* R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void quat_rot(const double* q, const double* v, double* o)
{
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__device__ __forceinline__ void quat_normalize(double* q)
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void quat_from_R(const double* R, double* q)      // Eigen::Quaterniond(Matrix3d)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        double qq[4]; qq[i] = 0.5 * t; t = 0.5 / t;
        qq[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; qq[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; qq[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
    }
}

__device__ __forceinline__ void huber(double e, double delta, double* rho)
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; }
}


// SE3Quat::exp(update) (se3quat.h:223-257), update = (omega, upsilon): quaternion eq (normalised) and translation et
__device__ __forceinline__ void se3_exp(const double* u, double* eq, double* et)
{
    const double om[3] = { u[0], u[1], u[2] }, up[3] = { u[3], u[4], u[5] };
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double O2[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) O2[a * 3 + c] = O[a * 3] * O[c] + O[a * 3 + 1] * O[3 + c] + O[a * 3 + 2] * O[6 + c];
    double R[9], V[9];
    if (theta < 0.00001) {
#pragma unroll
        for (int j = 0; j < 9; j++) { R[j] = ((j % 4) == 0 ? 1.0 : 0.0) + O[j] + O2[j]; V[j] = R[j]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
        for (int j = 0; j < 9; j++) { const double I = (j % 4) == 0 ? 1.0 : 0.0; R[j] = I + a * O[j] + b * O2[j]; V[j] = I + b * O[j] + c * O2[j]; }
    }
    quat_from_R(R, eq); quat_normalize(eq);
#pragma unroll
    for (int a = 0; a < 3; a++) et[a] = V[a * 3] * up[0] + V[a * 3 + 1] * up[1] + V[a * 3 + 2] * up[2];
}
// SE3Quat::operator* (se3quat.h:102-108): (q, t) <- (eq, et) * (q, t)
__device__ __forceinline__ void se3_premul(const double* eq, const double* et, double* q, double* t)
{
    double rt[3]; const double told[3] = { t[0], t[1], t[2] }; const double qold[4] = { q[0], q[1], q[2], q[3] };
    quat_rot(eq, told, rt);
    t[0] = et[0] + rt[0]; t[1] = et[1] + rt[1]; t[2] = et[2] + rt[2];
    double nq[4];
    nq[3] = eq[3] * qold[3] - eq[0] * qold[0] - eq[1] * qold[1] - eq[2] * qold[2];
    nq[0] = eq[3] * qold[0] + eq[0] * qold[3] + eq[1] * qold[2] - eq[2] * qold[1];
    nq[1] = eq[3] * qold[1] + eq[1] * qold[3] + eq[2] * qold[0] - eq[0] * qold[2];
    nq[2] = eq[3] * qold[2] + eq[2] * qold[3] + eq[0] * qold[1] - eq[1] * qold[0];
    quat_normalize(nq);
    q[0] = nq[0]; q[1] = nq[1]; q[2] = nq[2]; q[3] = nq[3];
}
