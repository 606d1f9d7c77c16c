// corb_map.cpp -- C-ABI host side of the map maintenance kernels (see include/corb_accel.h).  No CPU compute fallback.
#include "corb_internal.h"
#include "corb_workspace.h"
#include <vector>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
void corb_launch_distinctive(const unsigned long long* desc, const int* offset, int n_points, int* best_idx, int* status, hipStream_t s);
void corb_launch_rebase(const float* To2n, float* poses, int n_poses, float* points, int n_points, hipStream_t s);

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

extern "C" int corb_distinctive_descriptors(const uint8_t* desc, const int32_t* offset, int n_points, int32_t* best_idx, int device)
{
    if (n_points < 0 || !offset || !best_idx || (n_points > 0 && offset[n_points] > 0 && !desc)) { corb_set_error("corb_distinctive_descriptors: bad argument"); return CORB_ERR_ARG; }
    for (int p = 0; p < n_points; p++) if (offset[p + 1] < offset[p]) { corb_set_error("corb_distinctive_descriptors: offsets not ascending"); return CORB_ERR_ARG; }
    if (n_points == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const size_t total = (size_t)offset[n_points];
    CorbScratch scratch;
    unsigned long long* d_desc = nullptr; int *d_off = nullptr, *d_best = nullptr, *d_status = nullptr; int st = 0;
    HIPCHK(scratch.alloc(&d_desc, (total ? total : 1) * 4)); HIPCHK(scratch.alloc(&d_off, (size_t)n_points + 1));
    HIPCHK(scratch.alloc(&d_best, (size_t)n_points)); HIPCHK(scratch.alloc(&d_status, 1));
    // everything on the lane's own (non-blocking) stream: a null-stream memset / copy is not ordered against it
    if (total) HIPCHK(hipMemcpyAsync(d_desc, desc, total * 32, hipMemcpyHostToDevice, scratch.stream));
    HIPCHK(hipMemcpyAsync(d_off, offset, ((size_t)n_points + 1) * 4, hipMemcpyHostToDevice, scratch.stream));
    HIPCHK(hipMemsetAsync(d_status, 0, 4, scratch.stream));
    corb_launch_distinctive(d_desc, d_off, n_points, d_best, d_status, scratch.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best_idx, d_best, (size_t)n_points * 4, hipMemcpyDeviceToHost, scratch.stream));
    HIPCHK(hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, scratch.stream));
    HIPCHK(hipStreamSynchronize(scratch.stream));
    if (st) { corb_set_error("corb_distinctive_descriptors: a map point has more than 1024 observations"); return CORB_ERR_OVERFLOW; }
    return CORB_OK;
}

extern "C" int corb_rebase_map(const float* To2n, float* poses, int n_poses, float* points, int n_points, int device)
{
    if (!To2n || n_poses < 0 || n_points < 0 || (n_poses > 0 && !poses) || (n_points > 0 && !points)) { corb_set_error("corb_rebase_map: bad argument"); return CORB_ERR_ARG; }
    if (n_poses == 0 && n_points == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    CorbScratch scratch;
    float *d_T = nullptr, *d_poses = nullptr, *d_pts = nullptr;
    HIPCHK(scratch.alloc(&d_T, 16)); HIPCHK(scratch.alloc(&d_poses, (size_t)(n_poses ? n_poses : 1) * 16)); HIPCHK(scratch.alloc(&d_pts, (size_t)(n_points ? n_points : 1) * 3));
    HIPCHK(hipMemcpyAsync(d_T, To2n, 64, hipMemcpyHostToDevice, scratch.stream));
    if (n_poses) HIPCHK(hipMemcpyAsync(d_poses, poses, (size_t)n_poses * 64, hipMemcpyHostToDevice, scratch.stream));
    if (n_points) HIPCHK(hipMemcpyAsync(d_pts, points, (size_t)n_points * 12, hipMemcpyHostToDevice, scratch.stream));
    corb_launch_rebase(d_T, d_poses, n_poses, d_pts, n_points, scratch.stream);
    HIPCHK(hipGetLastError());
    if (n_poses) HIPCHK(hipMemcpyAsync(poses, d_poses, (size_t)n_poses * 64, hipMemcpyDeviceToHost, scratch.stream));
    if (n_points) HIPCHK(hipMemcpyAsync(points, d_pts, (size_t)n_points * 12, hipMemcpyDeviceToHost, scratch.stream));
    HIPCHK(hipStreamSynchronize(scratch.stream));
    return CORB_OK;
}
