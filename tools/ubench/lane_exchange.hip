// Checks every move of csrc/lane_exchange.h against its __shfl_xor / __shfl_up form on the device, bit for bit (tests/test_gpu_lane_exchange.py builds and runs it;
// add -DLX_USE_SWAP to check the v_permlane swap forms too).  Exit code = number of forms that differ.
#include "../../corb-slam_amd/csrc/lane_exchange.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define NF 20
__global__ void k(const double* in, const int* iin, double* out, long long* iout)
{
    const int lane = threadIdx.x, b = blockIdx.x;
    const double a = in[b * 128 + lane], c = in[b * 128 + 64 + lane];
    const int ia = iin[b * 128 + lane], ic = iin[b * 128 + 64 + lane];
    double* o = out + (size_t)b * NF * 128; long long* io = iout + (size_t)b * NF * 128;
#define PUT(f, x, y) do { o[(f) * 128 + lane] = (x); o[(f) * 128 + 64 + lane] = (y); } while (0)
#define IPUT(f, x, y) do { io[(f) * 128 + lane] = (long long)(x); io[(f) * 128 + 64 + lane] = (long long)(y); } while (0)
    PUT(0, lx_xor<1>(a), __shfl_xor(a, 1)); PUT(1, lx_xor<2>(a), __shfl_xor(a, 2)); PUT(2, lx_xor<4>(a), __shfl_xor(a, 4)); PUT(3, lx_xor<8>(a), __shfl_xor(a, 8));
    PUT(4, lx_xadd16(a, c), ((lane & 16) ? c : a) + __shfl_xor((lane & 16) ? a : c, 16));
    PUT(5, lx_xadd32(a, c), ((lane & 32) ? c : a) + __shfl_xor((lane & 32) ? a : c, 32));
    { double r = a; for (int s = 32; s > 0; s >>= 1) r += __shfl_xor(r, s); PUT(6, lx_wave_sum(a), r); }
    { double r = a; for (int s = 32; s > 0; s >>= 1) r = fmax(r, __shfl_xor(r, s)); PUT(7, lx_wave_max(a), r); }
    PUT(8, lx_add_xor<16>(a), a + __shfl_xor(a, 16)); PUT(9, lx_add_xor<4>(a), a + __shfl_xor(a, 4));
    IPUT(0, lx_xor_i<1>(ia), __shfl_xor(ia, 1)); IPUT(1, lx_xor_i<2>(ia), __shfl_xor(ia, 2)); IPUT(2, lx_xor_i<4>(ia), __shfl_xor(ia, 4)); IPUT(3, lx_xor_i<8>(ia), __shfl_xor(ia, 8));
    IPUT(4, lx_xadd16_i(ia, ic), ((lane & 16) ? ic : ia) + __shfl_xor((lane & 16) ? ia : ic, 16));
    IPUT(5, lx_xadd32_i(ia, ic), ((lane & 32) ? ic : ia) + __shfl_xor((lane & 32) ? ia : ic, 32));
    { int r = ia; for (int s = 32; s > 0; s >>= 1) r += __shfl_xor(r, s); IPUT(6, lx_wave_sum_i(ia), r); }
    { int r = ia; for (int s = 32; s > 0; s >>= 1) r = min(r, __shfl_xor(r, s)); IPUT(7, lx_wave_min_i(ia), r); }
    { int r = ia; for (int s = 32; s > 0; s >>= 1) r = max(r, __shfl_xor(r, s)); IPUT(8, lx_wave_max_i(ia), r); }
    { unsigned r = (unsigned)ia; for (int s = 32; s > 0; s >>= 1) r = min(r, (unsigned)__shfl_xor((int)r, s)); IPUT(9, lx_wave_min_u((unsigned)ia), r); }
    { unsigned long long v = ((unsigned long long)(unsigned)ia << 32) | (unsigned)ic, r = v; for (int s = 32; s > 0; s >>= 1) { const unsigned long long t = __shfl_xor(r, s); r = t < r ? t : r; }
      IPUT(10, lx_wave_min_u64(v), r); r = v; for (int s = 32; s > 0; s >>= 1) { const unsigned long long t = __shfl_xor(r, s); r = t > r ? t : r; } IPUT(11, lx_wave_max_u64(v), r); }
    { int r = ia & 1023; const int v0 = r; for (int s = 1; s < 64; s <<= 1) { const int t = __shfl_up(r, s); if (lane >= s) r += t; } IPUT(12, lx_wave_incl_scan_i(v0), r); }
}
int main()
{
    const int NB = 32;
    static double h[NB * 128], o[NB * NF * 128]; static int ih[NB * 128]; static long long io[NB * NF * 128];
    srand(7);
    for (int i = 0; i < NB * 128; i++) { h[i] = ((double)rand() / RAND_MAX - 0.3) * (1 + i % 9); ih[i] = rand() - RAND_MAX / 3; }
    double *di, *dout; int* dii; long long* diout;
    (void)hipMalloc(&di, sizeof(h)); (void)hipMalloc(&dout, sizeof(o)); (void)hipMalloc(&dii, sizeof(ih)); (void)hipMalloc(&diout, sizeof(io));
    (void)hipMemset(dout, 0, sizeof(o)); (void)hipMemset(diout, 0, sizeof(io));
    (void)hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice); (void)hipMemcpy(dii, ih, sizeof(ih), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, di, dii, dout, diout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); (void)hipMemcpy(io, diout, sizeof(io), hipMemcpyDeviceToHost);
    const char* nm[10] = {"xor1", "xor2", "xor4", "xor8", "xadd16", "xadd32", "wave_sum", "wave_max", "add_xor16", "add_xor4"};
    const char* inm[13] = {"xor1_i", "xor2_i", "xor4_i", "xor8_i", "xadd16_i", "xadd32_i", "wave_sum_i", "wave_min_i", "wave_max_i", "wave_min_u", "wave_min_u64", "wave_max_u64", "incl_scan_i"};
    int bad = 0;
    for (int f = 0; f < 10; f++) { int d = 0; for (int b = 0; b < NB; b++) d += memcmp(o + ((size_t)b * NF + f) * 128, o + ((size_t)b * NF + f) * 128 + 64, 64 * 8) != 0; printf("%-12s %s\n", nm[f], d ? "DIFFERENT" : "equal"); bad += d != 0; }
    for (int f = 0; f < 13; f++) { int d = 0; for (int b = 0; b < NB; b++) d += memcmp(io + ((size_t)b * NF + f) * 128, io + ((size_t)b * NF + f) * 128 + 64, 64 * 8) != 0; printf("%-12s %s\n", inm[f], d ? "DIFFERENT" : "equal"); bad += d != 0; }
    return bad;
}
