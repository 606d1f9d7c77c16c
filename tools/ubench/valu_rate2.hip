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
            if (OP == 0) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 2) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x) :: "vcc", "s20");
            if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 4) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x) :: "vcc", "s20");
            if (OP == 5) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 6) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 13) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 14) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(z));
            if (OP == 15) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 16) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 17) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 19) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x) :: "vcc", "s20");
            if (OP == 20) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 21) asm volatile("v_max_u16 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 22) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 23) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 24) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 25) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 27) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 28) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 29) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 30) asm volatile("v_readfirstlane_b32 s20, %0" : "+v"(x) :: "vcc", "s20");
            if (OP == 31) asm volatile("v_rndne_f32 %0, %0" : "+v"(x) :: "vcc", "s20");
            if (OP == 32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x) :: "vcc", "s20");
            if (OP == 33) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 34) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y) : "v"(z), "v"(z2));
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
    run<0>("v_min_i32", out, cyc);
    run<1>("v_and_b32", out, cyc);
    run<2>("v_lshlrev_b32", out, cyc);
    run<3>("v_cndmask_b32", out, cyc);
    run<4>("v_bfe_u32", out, cyc);
    run<5>("v_add3_u32", out, cyc);
    run<6>("v_lshl_or_b32", out, cyc);
    run<7>("v_lshl_add_u32", out, cyc);
    run<8>("v_mul_u32_u24", out, cyc);
    run<9>("v_mul_f32", out, cyc);
    run<10>("v_fmac_f32", out, cyc);
    run<11>("v_pk_max_i16", out, cyc);
    run<12>("v_pk_min_f16", out, cyc);
    run<13>("v_pk_mad_u16", out, cyc);
    run<14>("v_pk_mul_f32", out, cyc);
    run<15>("v_sad_u16", out, cyc);
    run<16>("v_alignbyte_b32", out, cyc);
    run<17>("v_mad_u32_u16", out, cyc);
    run<18>("v_med3_i32", out, cyc);
    run<19>("v_cvt_f32_ubyte1", out, cyc);
    run<20>("v_mov_dpp", out, cyc);
    run<21>("v_max_u16", out, cyc);
    run<22>("v_min3_f32", out, cyc);
    run<23>("v_xor3", out, cyc);
    run<24>("v_bfi_b32", out, cyc);
    run<25>("v_or3_b32", out, cyc);
    run<26>("v_sub_u32", out, cyc);
    run<27>("v_mul_i32_i24", out, cyc);
    run<28>("v_mad_i32_i24", out, cyc);
    run<29>("v_cmp_lt_u32", out, cyc);
    run<30>("v_readlane", out, cyc);
    run<31>("v_rndne_f32", out, cyc);
    run<32>("v_cvt_i32_f32", out, cyc);
    run<33>("v_add_f32", out, cyc);
    run<34>("v_fma_f64", out, cyc);
    return 0;
}
