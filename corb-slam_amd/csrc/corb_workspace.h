// corb_workspace.h -- per-device scratch workspace shared by every host-pointer entry point of the C-ABI (matchers, BA, pose / Sim3 /
// essential-graph optimisation, map maintenance).  Host code only.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <initializer_list>
#include <cstring>
#include <mutex>
#include <algorithm>

// Per-device workspace that lives as long as the process: stream, timing events and a bump arena of device
// memory.  A BA call used to pay ~45 hipMalloc/hipFree, a stream and six events (several ms -- more than the
// whole optimisation of a local window); now it takes the workspace (one call at a time per device), bumps pointers, and
// resets the arena on exit.  The arena grows by chunks; after a call that needed several, they are merged into one.
struct CorbWorkspace {
    std::mutex mu;
    hipStream_t stream = nullptr; hipEvent_t ev[10] = {};
    void* pinned = nullptr;           // 4 KB of page-locked host memory: read-backs of a few scalars that must not block the host inside hipMemcpyAsync
    struct Chunk { char* base; size_t cap, used; };
    std::vector<Chunk> chunks;
    // page-locked staging for the small uploads / read-backs of the per-frame calls: a copy out of (into) pageable memory is a synchronous ~25-30 us affair
    // in the runtime; out of pinned memory it is an asynchronous enqueue behind which the kernels are launched at once.  Bump-allocated per call, grown
    // between calls to what the largest call wanted (up to 32 MB: larger transfers go the direct way).
    char* hstage = nullptr; size_t hcap = 0, hused = 0, hwant = 0;
    void* host_take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        hwant = std::max(hwant, hused + bytes);
        if (hused + bytes > hcap) return nullptr;
        void* p = hstage + hused; hused += bytes; return p;
    }
    hipError_t ensure() {
        if (stream) return hipSuccess;
        hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking); if (e != hipSuccess) return e;
        for (auto& v : ev) { e = hipEventCreate(&v); if (e != hipSuccess) return e; }
        return hipHostMalloc(&pinned, 4096);
    }
    hipError_t take(void** out, size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255; if (!bytes) bytes = 256;
        for (auto& c : chunks) if (c.cap - c.used >= bytes) { *out = c.base + c.used; c.used += bytes; return hipSuccess; }
        size_t total = 0; for (auto& c : chunks) total += c.cap;
        const size_t cap = std::max(bytes, std::max(total, (size_t)8 << 20));           // at least double
        void* p = nullptr; hipError_t e = hipMalloc(&p, cap); if (e != hipSuccess) return e;
        chunks.push_back({(char*)p, cap, bytes}); *out = p; return hipSuccess;
    }
    void reset() {
        hused = 0;
        if (hwant > hcap && hwant <= ((size_t)32 << 20)) {
            if (hstage) (void)hipHostFree(hstage);
            hcap = std::max(hwant + hwant / 2, (size_t)1 << 20); hstage = nullptr;
            if (hipHostMalloc((void**)&hstage, hcap) != hipSuccess) { hstage = nullptr; hcap = 0; }
        }
        hwant = 0;
        if (chunks.size() > 1) {                       // merge: next call finds one chunk that holds everything
            size_t total = 0; for (auto& c : chunks) { total += c.cap; (void)hipFree(c.base); }
            chunks.clear();
            void* p = nullptr; if (hipMalloc(&p, total) == hipSuccess) chunks.push_back({(char*)p, total, 0});
        } else for (auto& c : chunks) c.used = 0;
    }
};
// Two lanes per device so that a long optimisation (global / local bundle adjustment, essential graph: lane 1) never blocks the
// short per-frame calls of the tracking thread (matchers, pose / Sim3 optimisation, map maintenance: lane 0) on the workspace mutex.
inline CorbWorkspace& corb_workspace(int device, int lane) { static CorbWorkspace ws[64][2]; return ws[device < 0 || device >= 64 ? 0 : device][lane ? 1 : 0]; }

struct CorbScratch {                         // one BA call's view of the workspace: everything taken is released on scope exit
    CorbWorkspace* ws = nullptr; std::unique_lock<std::mutex> lock;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> evs;      // (events are the workspace's: nothing to destroy)
    explicit CorbScratch(int lane = 0) {
        int dev = 0; (void)hipGetDevice(&dev);
        ws = &corb_workspace(dev, lane); lock = std::unique_lock<std::mutex>(ws->mu);
        if (ws->ensure() == hipSuccess) stream = ws->stream;
    }
    ~CorbScratch() { if (stream) (void)hipStreamSynchronize(stream); ws->reset(); }   // this call's work only: everything is issued on the lane's own (non-blocking) stream
    hipEvent_t event(int i) { return ws->ev[i]; }
    void* pinned() { return ws->pinned; }
    template <class T> hipError_t alloc(T** out, size_t n) { void* p = nullptr; hipError_t e = ws->take(&p, (n ? n : 1) * sizeof(T)); if (e == hipSuccess) *out = (T*)p; return e; }
    // host -> device: through the pinned staging when it has room (asynchronous on the lane's stream; the staging is released when the call ends), else directly
    hipError_t h2d(void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        if (void* st = ws->host_take(bytes)) { memcpy(st, src, bytes); return hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, stream); }
        return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    }
    // device -> host results: enqueue into the staging, copy out after the caller's synchronisation (fetch_finish); falls back to a direct copy
    struct Pending { void* dst; const void* st; size_t bytes; };
    std::vector<Pending> pending;
    hipError_t d2h(void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        if (void* st = ws->host_take(bytes)) { pending.push_back({dst, st, bytes}); return hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, stream); }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    }
    hipError_t fetch_finish() {          // synchronises the lane's stream and hands the staged results over
        hipError_t e = hipStreamSynchronize(stream);
        if (e == hipSuccess) for (const Pending& p : pending) memcpy(p.dst, p.st, p.bytes);
        pending.clear();
        return e;
    }
    template <class T> hipError_t upload(T** out, const T* src, size_t n) { hipError_t e = alloc(out, n); if (e == hipSuccess && n) e = h2d(*out, src, n * sizeof(T)); return e; }
    // several small host arrays as ONE allocation and ONE copy (a synchronous copy of a few KB costs ~15 us each; per-frame calls upload up to a dozen)
    struct Piece { void** dst; const void* src; size_t bytes; };
    hipError_t upload_block(std::initializer_list<Piece> pieces) {
        size_t total = 0;
        for (const Piece& pc : pieces) total += (pc.bytes + 255) & ~(size_t)255;
        char* base = nullptr; hipError_t e = alloc(&base, total + 256); if (e != hipSuccess) return e;
        size_t off = 0;
        if (char* st = static_cast<char*>(ws->host_take(total))) {
            for (const Piece& pc : pieces) { if (pc.bytes) memcpy(st + off, pc.src, pc.bytes); *pc.dst = base + off; off += (pc.bytes + 255) & ~(size_t)255; }
            return total ? hipMemcpyAsync(base, st, total, hipMemcpyHostToDevice, stream) : hipSuccess;
        }
        static thread_local std::vector<char> blob;
        blob.resize(total + 1);
        for (const Piece& pc : pieces) { if (pc.bytes) memcpy(blob.data() + off, pc.src, pc.bytes); *pc.dst = base + off; off += (pc.bytes + 255) & ~(size_t)255; }
        return total ? hipMemcpy(base, blob.data(), total, hipMemcpyHostToDevice) : hipSuccess;
    }
    template <class T> hipError_t upload(T** out, const std::vector<T>& v) { hipError_t e = alloc(out, v.size()); if (e == hipSuccess && !v.empty()) e = h2d(*out, v.data(), v.size() * sizeof(T)); return e; }
};

