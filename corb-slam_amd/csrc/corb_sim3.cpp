// corb_sim3.cpp -- C-ABI host side of Optimizer::OptimizeSim3 (see include/corb_accel.h).  No CPU compute fallback.
#include "sim3_internal.h"
#include "corb_workspace.h"
#include <vector>
#include <cmath>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);


namespace {
using DevPool = CorbScratch;
// Eigen::Quaterniond(Matrix3d): the g2o::Sim3(R, t, s) constructor; NOT normalised afterwards
void quat_from_R(const double* R, double* q)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
void quat_to_R(const double* q, double* R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
}  // namespace

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

extern "C" int corb_optimize_sim3(const CorbSim3Problem* problems, int n_problems, double* R12, double* t12, double* s12, float th2, int fix_scale,
                                  uint8_t* const* removed, int32_t* n_inliers, int32_t* iterations, int device)
{
    if (!problems || n_problems < 1 || !R12 || !t12 || !s12 || !n_inliers || !(th2 > 0)) { corb_set_error("corb_optimize_sim3: bad argument"); return CORB_ERR_ARG; }
    std::vector<int> off(1, 0);
    std::vector<float> p1, p2, o1, o2, w1, w2, K;
    std::vector<double> S((size_t)8 * n_problems);
    for (int f = 0; f < n_problems; f++) {
        const CorbSim3Problem& P = problems[f];
        if (P.n < 0 || (P.n > 0 && (!P.p1c || !P.p2c || !P.obs1 || !P.obs2 || !P.inv_sigma2_1 || !P.inv_sigma2_2))) { corb_set_error("corb_optimize_sim3: problem %d: bad argument", f); return CORB_ERR_ARG; }
        p1.insert(p1.end(), P.p1c, P.p1c + 3 * (size_t)P.n); p2.insert(p2.end(), P.p2c, P.p2c + 3 * (size_t)P.n);
        o1.insert(o1.end(), P.obs1, P.obs1 + 2 * (size_t)P.n); o2.insert(o2.end(), P.obs2, P.obs2 + 2 * (size_t)P.n);
        w1.insert(w1.end(), P.inv_sigma2_1, P.inv_sigma2_1 + P.n); w2.insert(w2.end(), P.inv_sigma2_2, P.inv_sigma2_2 + P.n);
        const float k[8] = { P.fx1, P.fy1, P.cx1, P.cy1, P.fx2, P.fy2, P.cx2, P.cy2 };
        K.insert(K.end(), k, k + 8);
        off.push_back(off.back() + P.n);
        quat_from_R(R12 + 9 * (size_t)f, &S[8 * (size_t)f]);
        for (int a = 0; a < 3; a++) S[8 * (size_t)f + 4 + a] = t12[3 * (size_t)f + a];
        S[8 * (size_t)f + 7] = s12[f];
    }
    int rc = corb_select_device(device); if (rc) return rc;
    const size_t N = (size_t)off.back();
    DevPool pool;
    CorbSim3Dev d; memset(&d, 0, sizeof(d));
    int* doff; float *dp1, *dp2, *do1, *do2, *dw1, *dw2, *dK; double *dS, *dl12, *dl21; unsigned char* drem; int* dcnt;
    HIPCHK(pool.upload_block({{(void**)&doff, off.data(), off.size() * sizeof(off[0])}, {(void**)&dp1, p1.data(), p1.size() * sizeof(p1[0])}, {(void**)&dp2, p2.data(), p2.size() * sizeof(p2[0])},
                              {(void**)&do1, o1.data(), o1.size() * sizeof(o1[0])}, {(void**)&do2, o2.data(), o2.size() * sizeof(o2[0])}, {(void**)&dw1, w1.data(), w1.size() * sizeof(w1[0])},
                              {(void**)&dw2, w2.data(), w2.size() * sizeof(w2[0])}, {(void**)&dK, K.data(), K.size() * sizeof(K[0])}, {(void**)&dS, S.data(), S.size() * sizeof(S[0])}}));
    HIPCHK(pool.alloc(&dl12, N)); HIPCHK(pool.alloc(&dl21, N)); HIPCHK(pool.alloc(&drem, N)); HIPCHK(pool.alloc(&dcnt, (size_t)4 * n_problems));
    d.n_problems = n_problems; d.off = doff; d.p1c = dp1; d.p2c = dp2; d.obs1 = do1; d.obs2 = do2; d.w1 = dw1; d.w2 = dw2; d.K = dK; d.S = dS;
    d.removed = drem; d.last12 = dl12; d.last21 = dl21; d.counters = dcnt; d.th2 = th2; d.fix_scale = fix_scale ? 1 : 0;
    sim3_launch_optimize(d, pool.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(pool.stream));
    std::vector<unsigned char> rem(N ? N : 1); std::vector<int> cnt((size_t)4 * n_problems);
    HIPCHK(hipMemcpy(S.data(), dS, S.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (N) HIPCHK(hipMemcpy(rem.data(), drem, N, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cnt.data(), dcnt, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int f = 0; f < n_problems; f++) {
        n_inliers[f] = cnt[4 * (size_t)f + 2];
        if (iterations) iterations[f] = cnt[4 * (size_t)f];
        if (cnt[4 * (size_t)f + 3]) {                           // g2oS12 = vSim3_recov->estimate() (only when the second round ran)
            quat_to_R(&S[8 * (size_t)f], R12 + 9 * (size_t)f);
            for (int a = 0; a < 3; a++) t12[3 * (size_t)f + a] = S[8 * (size_t)f + 4 + a];
            s12[f] = S[8 * (size_t)f + 7];
        }
        if (removed && removed[f]) for (int i = 0; i < problems[f].n; i++) removed[f][i] = rem[off[f] + i];
    }
    return CORB_OK;
}
