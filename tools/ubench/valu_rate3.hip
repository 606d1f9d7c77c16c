// Micro-benchmark (development aid, not part of the product): VALU issue rate on gfx950 by instruction type and
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_INNER 64
#define N_OUTER 2048
template <int OP> __global__ void k(uint32_t* out, uint64_t* cyc, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t b = seed ^ 0x1234, c = seed + 99; double dd[4] = {1.0 + seed, 2.0, 3.0, 4.0}; double z = 1.0000001, z2 = 0.5;
    const uint64_t t0 = clock64();
    for (int o = 0; o < N_OUTER; o++) {
#pragma unroll
        for (int i = 0; i < N_INNER; i++) {
            uint32_t& x = a[i & 7]; double& y = dd[i & 3];
            if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x) :: "vcc");
            if (OP == 2) asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 3) asm volatile("v_floor_f32 %0, %0" : "+v"(x) :: "vcc");
            if (OP == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 5) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 7) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x) :: "vcc");
            if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 9) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 11) asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 15) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x) :: "vcc");
            if (OP == 16) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x) :: "vcc");
            if (OP == 17) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 18) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(z));
        }
    }
    const uint64_t t1 = clock64();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += (uint32_t)dd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* out, uint64_t* cyc)
{
    for (int wps = 4; wps <= 4; wps *= 2) {                 // waves per SIMD: block = 256 threads = 1 wave per SIMD; wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, cyc, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, cyc, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr_per_simd = (double)wps * N_INNER * N_OUTER;
        printf("%-22s waves/SIMD %d  %.3f ms  %.2f ns/instr/SIMD  s_memtime cycles/instr/SIMD %.2f (clock64 ticks %llu)\n", name, wps, ms, ms * 1e6 / instr_per_simd,
               (double)c / instr_per_simd, (unsigned long long)c);
    }
}
int main()
{
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("v_min_f32", out, cyc);
    run<1>("v_cvt_u32_f32", out, cyc);
    run<2>("v_cvt_pk_u8_f32", out, cyc);
    run<3>("v_floor_f32", out, cyc);
    run<4>("v_min_u32", out, cyc);
    run<5>("v_or_b32", out, cyc);
    run<6>("v_xor_b32", out, cyc);
    run<7>("v_lshrrev_b32", out, cyc);
    run<8>("v_mov_b32", out, cyc);
    run<9>("v_min_u16", out, cyc);
    run<10>("v_add_u16", out, cyc);
    run<11>("v_mul_lo_u16", out, cyc);
    run<12>("v_mad_u16", out, cyc);
    run<13>("v_max_f32", out, cyc);
    run<14>("v_mac_f32/fmac", out, cyc);
    run<15>("v_cvt_f32_u32", out, cyc);
    run<16>("v_cvt_f32_ubyte0", out, cyc);
    run<17>("v_add_co_u32", out, cyc);
    run<18>("v_subrev_u32", out, cyc);
    run<19>("v_pk_fma_f32", out, cyc);
    return 0;
}
