// checks lx_wave_incl_scan_i (csrc/lane_exchange.h) against the __shfl_up form on the device
#include "../../corb-slam_amd/csrc/lane_exchange.h"
#include <cstdio>
#include <cstdlib>
__global__ void k(const int* in, int* out)
{
    const int lane = threadIdx.x;
    int v = in[blockIdx.x * 64 + lane], r = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(r, o); if (lane >= o) r += t; }
    out[blockIdx.x * 128 + lane] = lx_wave_incl_scan_i(v);
    out[blockIdx.x * 128 + 64 + lane] = r;
}
int main()
{
    const int NB = 64;
    int h[NB * 64], o[NB * 128]; for (int i = 0; i < NB * 64; i++) h[i] = rand() % 1000 - 300;
    int *di, *dout; (void)hipMalloc(&di, sizeof(h)); (void)hipMalloc(&dout, sizeof(o)); (void)hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, di, dout); (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; b++) for (int l = 0; l < 64; l++) if (o[b * 128 + l] != o[b * 128 + 64 + l]) bad++;
    printf("scan mismatches %d\n", bad);
    return bad != 0;
}
