// Micro-benchmark (development aid, not part of the product): FP64 matrix-core facts on gfx950 that the guides do not state.
//  1. lane -> element layout of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 products per instruction), probed with one-hot inputs;
//  2. issue rate of v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and v_fma_f64 per SIMD (=> the FP64 roofs the BA kernels are priced against).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

// out[la][lb][lane] = D value of `lane` when A is one-hot at lane la (value 1) and B one-hot at lane lb (value 1)
__global__ void probe(float* out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = (float)d;
        }
}

template <int OP> __global__ void rate(double* out, int n_outer)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    for (int o = 0; o < n_outer; o++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                           c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); }
            if (OP == 1) { s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
                           s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0); }
            if (OP == 2) { s0 = fma(a, b, s0); s1 = fma(a, b, s1); s2 = fma(a, b, s2); s3 = fma(a, b, s3); s4 = fma(a, b, s4); s5 = fma(a, b, s5); s6 = fma(a, b, s6); s7 = fma(a, b, s7); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
template <int OP> void run(const char* name, double* out, double flop_per_instr, int instr_per_inner)
{
    const int n_outer = 4096;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        rate<OP><<<blocks, 256>>>(out, 16); hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<OP><<<blocks, 256>>>(out, n_outer);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * n_outer * 16 * instr_per_inner;          // wave instructions in total
        printf("%-28s waves/SIMD %d  %.3f ms  %.2f ns/instr/SIMD  %.1f TFLOP/s\n", name, wps, ms, ms * 1e6 / (instr / 1024.0), instr * flop_per_instr / (ms * 1e-3) * 1e-12);
    }
}
int main()
{
    float* dp; hipMalloc(&dp, 64 * 64 * 64 * 4);
    probe<<<1, 64>>>(dp);
    std::vector<float> h(64 * 64 * 64);
    hipMemcpy(h.data(), dp, h.size() * 4, hipMemcpyDeviceToHost);
    // derive: for every output lane, which (la, lb) pairs contribute -> block/row/col/k coordinates
    printf("# v_mfma_f64_4x4x4_4b_f64 layout probe: out_lane <- list of (a_lane, b_lane)\n");
    for (int lane = 0; lane < 64; lane++) {
        printf("D lane %2d:", lane);
        for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) if (h[(la * 64 + lb) * 64 + lane] != 0.0f) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    double* out; hipMalloc(&out, 256 * 4 * 256 * 8);
    run<0>("v_mfma_f64_16x16x4_f64", out, 2.0 * 16 * 16 * 4, 4);
    run<1>("v_mfma_f64_4x4x4_4b_f64", out, 2.0 * 4 * 4 * 4 * 4, 4);
    run<2>("v_fma_f64", out, 2.0 * 64, 8);
    return 0;
}
