// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: streams of a KNOWN byte count, far beyond the 256 MiB
// Infinity Cache, read / written with 1, 4, 8 and 16 bytes per lane, and 144-byte blocks gathered in 16-byte pieces (the V blocks of the Schur kernels).
// build: hipcc --offload-arch=gfx950 -O3 -w fetch_calib.hip -o fetch_calib ;  run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (tools/gpu_fetch_calib.sh); every kernel prints the bytes it moves per launch, the script divides the counters by them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> __global__ __launch_bounds__(256) void calib_read(const T* __restrict__ in, size_t n, unsigned long long* out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const T v = in[i]; const unsigned char* b = reinterpret_cast<const unsigned char*>(&v); acc += b[0]; }
    if (acc == 0x123456789ull) *out = acc;                 // (keeps the loads alive)
}
template <class T> __global__ __launch_bounds__(256) void calib_write(T* __restrict__ outp, size_t n, int seed)
{
    T v; unsigned char* b = reinterpret_cast<unsigned char*>(&v);
    for (int k = 0; k < (int)sizeof(T); k++) b[k] = (unsigned char)(seed + k);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) outp[i] = v;
}
// 144-byte blocks in a pseudo-random order, nine 16-byte pieces each, consecutive lanes on consecutive pieces (the row kernel's staging pattern)
__global__ __launch_bounds__(256) void calib_gather144(const uint4* __restrict__ in, size_t nblocks, unsigned long long* out)
{
    unsigned long long acc = 0;
    const size_t npieces = nblocks * 9;
    for (size_t m = (size_t)blockIdx.x * 256 + threadIdx.x; m < npieces; m += (size_t)gridDim.x * 256) {
        const size_t blk = m / 9, part = m - 9 * blk;
        const size_t src = (blk * 2654435761ull) % nblocks;       // a permutation-like scatter of the block index
        acc += in[src * 9 + part].x;
    }
    if (acc == 0x123456789ull) *out = acc;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;                  // 2 GiB per stream: 8x the Infinity Cache
    void* buf = nullptr; unsigned long long* out = nullptr;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&out, 8)); CHK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; rep++) {
        calib_read<unsigned char><<<grid, 256>>>((const unsigned char*)buf, bytes / 4, out);          // (a quarter of the buffer: 1-byte loads are slow)
        calib_read<uint32_t><<<grid, 256>>>((const uint32_t*)buf, bytes / 4, out);
        calib_read<uint2><<<grid, 256>>>((const uint2*)buf, bytes / 8, out);
        calib_read<uint4><<<grid, 256>>>((const uint4*)buf, bytes / 16, out);
        calib_gather144<<<grid, 256>>>((const uint4*)buf, bytes / 144, out);
        calib_write<unsigned char><<<grid, 256>>>((unsigned char*)buf, bytes / 4, rep);
        calib_write<uint32_t><<<grid, 256>>>((uint32_t*)buf, bytes / 4, rep);
        calib_write<uint2><<<grid, 256>>>((uint2*)buf, bytes / 8, rep);
        calib_write<uint4><<<grid, 256>>>((uint4*)buf, bytes / 16, rep);
    }
    CHK(hipDeviceSynchronize());
    printf("bytes_per_launch calib_read<unsigned char> %zu\n", bytes / 4);
    printf("bytes_per_launch calib_read<unsigned int> %zu\n", bytes);
    printf("bytes_per_launch calib_read<HIP_vector_type<unsigned int, 2u> > %zu\n", bytes);
    printf("bytes_per_launch calib_read<HIP_vector_type<unsigned int, 4u> > %zu\n", bytes);
    printf("bytes_per_launch calib_gather144 %zu\n", bytes / 144 * 144);
    printf("bytes_per_launch calib_write<unsigned char> %zu\n", bytes / 4);
    printf("bytes_per_launch calib_write<unsigned int> %zu\n", bytes);
    printf("bytes_per_launch calib_write<HIP_vector_type<unsigned int, 2u> > %zu\n", bytes);
    printf("bytes_per_launch calib_write<HIP_vector_type<unsigned int, 4u> > %zu\n", bytes);
    return 0;
}
