#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double xchg_add32(double a, double b)   // lower lanes: a(own) + a(lane+32); upper lanes: b(own) + b(lane-32)
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    v2u r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    v2u r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double(r1.x, r0.x) + __hiloint2double(r1.y, r0.y);
}
__device__ __forceinline__ double xchg_add16(double a, double b)
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    v2u r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    v2u r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double(r1.x, r0.x) + __hiloint2double(r1.y, r0.y);
}
template <int CTRL, int BANK> __device__ __forceinline__ double dpp_mov(double old, double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor8(double v) { return dpp_mov<0x128, 0xF>(v, v); }          // row_ror:8
__device__ __forceinline__ double xor4(double v) { double t = dpp_mov<0x104, 0x5>(v, v); return dpp_mov<0x114, 0xA>(t, v); }   // row_shl:4 into banks 0,2 ; row_shr:4 into banks 1,3
__device__ __forceinline__ double xor2(double v) { return dpp_mov<0x4E, 0xF>(v, v); }           // quad_perm [2,3,0,1]
__device__ __forceinline__ double xor1(double v) { return dpp_mov<0xB1, 0xF>(v, v); }           // quad_perm [1,0,3,2]
__global__ void k(const double* in, double* out)
{
    const int lane = threadIdx.x;
    double a = in[lane], b = in[64 + lane];
    out[lane] = xchg_add32(a, b);
    out[64 + lane] = ((lane & 32) ? b : a) + __shfl_xor((lane & 32) ? a : b, 32);
    out[128 + lane] = xchg_add16(a, b);
    out[192 + lane] = ((lane & 16) ? b : a) + __shfl_xor((lane & 16) ? a : b, 16);
    out[256 + lane] = xor8(a);  out[320 + lane] = __shfl_xor(a, 8);
    out[384 + lane] = xor4(a);  out[448 + lane] = __shfl_xor(a, 4);
    out[512 + lane] = xor2(a);  out[576 + lane] = __shfl_xor(a, 2);
    out[640 + lane] = xor1(a);  out[704 + lane] = __shfl_xor(a, 1);
}
int main()
{
    double h[128], o[768]; for (int i = 0; i < 128; i++) h[i] = (double)rand() / RAND_MAX + i;
    double *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout); hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    const char* nm[6] = {"swap32", "swap16", "xor8", "xor4", "xor2", "xor1"};
    int bad = 0;
    for (int t = 0; t < 6; t++) { int ok = memcmp(o + 128 * t, o + 128 * t + 64, 64 * 8) == 0; printf("%s %s\n", nm[t], ok ? "equal" : "DIFFERENT"); bad += !ok; }
    return bad;
}
