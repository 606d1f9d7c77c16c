// corb_comm.cpp -- the client -> server map push (see include/corb_accel.h): communicator over RCCL (one process per GPU, xGMI) or over an in-process
// transport, the push's bookkeeping as a pure function (corb_map_push_plan), and the collective-safe push itself.
// Replaces, for the hot path, the boost-text-archive service batches of corbslam_client/src/Cache.cc:322-375 / DataDriver.cc:135-193 and their server
// side corbslam_server/src/MapFusion.cpp:31-190: a push is a handful of messages of whole records between the ranks' device buffers.
#include "store_host.h"
#include "corb_workspace.h"
#include <dlfcn.h>
#include <string>
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

// ---- RCCL (librccl.so loaded on first use: a process that never pushes a map does not pay for it) ----
namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int version = 0;
    bool ok = false;
    std::string why;
};
Rccl& rccl()
{
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        // the RCCL that belongs to the HIP runtime THIS library is linked with (its directory): a process may hold a second copy of the ROCm libraries
        // (a Python framework's bundled ones), and streams / events of one runtime mean nothing to the other
        std::string own;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            own = info.dli_fname;
            const size_t slash = own.rfind('/');
            own = slash == std::string::npos ? std::string() : own.substr(0, slash + 1) + "librccl.so";
        }
        for (const char* name : {own.c_str(), "/opt/rocm/lib/librccl.so", "librccl.so", "librccl.so.1"}) { if (!*name) continue; r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) { r.why = "librccl.so could not be loaded"; return; }
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetVersion = (int (*)(int*))sym("ncclGetVersion");
        r.GetUniqueId = (int (*)(ncclUniqueId_*))sym("ncclGetUniqueId"); r.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId_, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy"); r.Send = (int (*)(const void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
        r.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
        r.GroupStart = (int (*)())sym("ncclGroupStart"); r.GroupEnd = (int (*)())sym("ncclGroupEnd"); r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        const bool syms = r.GetVersion && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllGather && r.GroupStart && r.GroupEnd;
        if (!syms) { r.why = "librccl.so lacks a symbol this library binds"; return; }
        // The prototypes and enum constants below are declared by hand (no rccl.h at build time): they are those of the NCCL 2.x API -- ncclInt8 = 0,
        // ncclInt32 = 2, a 128-byte ncclUniqueId passed by value, ncclSend / ncclRecv (2.7+).  Refuse any other major version instead of drifting silently.
        if (r.GetVersion(&r.version) != 0 || r.version < 20700 || r.version >= 30000) {
            r.why = "librccl.so reports version " + std::to_string(r.version) + "; this library binds the NCCL 2.7 .. 2.x API"; return;
        }
        r.ok = true;
    });
    return r;
}
const int NCCL_INT8 = 0, NCCL_INT32 = 2;      // ncclDataType_t: ncclInt8 = 0 (= ncclChar), ncclInt32 = 2 (rccl.h, NCCL 2.x)
}
#define NCCLCHK(call) do { int e_ = (call); if (e_ != 0) { corb_set_error("%s failed: %s", #call, rccl().GetErrorString ? rccl().GetErrorString(e_) : "rccl error"); return CORB_ERR_HIP; } } while (0)

// ---- transports ----
namespace {
struct Msg { void* ptr; size_t bytes; int peer; };

// in-process transport: `world` communicators share one hub; every rank is driven by its own host thread
struct LocalHub {
    int world = 1;
    std::mutex mu; std::condition_variable cv;
    int arrived = 0; unsigned long long generation = 0;
    std::vector<int> table;                              // all-gather staging
    std::vector<std::vector<Msg>> posted;                // posted[sender] = the sends of the current exchange (peer = receiver)
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long g = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};
}

struct CorbComm {
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    // RCCL
    ncclComm_t comm = nullptr; int* d_ints = nullptr; int d_ints_cap = 0;
    // in-process
    std::shared_ptr<LocalHub> hub;

    // every rank contributes n ints; all[r * n + k] = rank r's k-th
    int all_gather(const int* mine, int n, int* all) {
        if (hub) {
            { std::lock_guard<std::mutex> lk(hub->mu); if ((int)hub->table.size() < world * n) hub->table.resize((size_t)world * n); }
            hub->barrier();                                              // (the table has its size on every rank's view)
            memcpy(&hub->table[(size_t)rank * n], mine, sizeof(int) * n);
            hub->barrier();
            memcpy(all, hub->table.data(), sizeof(int) * (size_t)world * n);
            hub->barrier();                                              // nobody overwrites the table while another rank still reads it
            return CORB_OK;
        }
        if (n * (world + 1) > d_ints_cap) { corb_set_error("corb_comm: all-gather of %d ints per rank exceeds the staging buffer", n); return CORB_ERR_ARG; }
        HIPCHK(hipMemcpyAsync(d_ints + (size_t)world * n, mine, sizeof(int) * n, hipMemcpyHostToDevice, stream));
        NCCLCHK(rccl().AllGather(d_ints + (size_t)world * n, d_ints, (size_t)n, NCCL_INT32, comm, stream));
        HIPCHK(hipMemcpyAsync(all, d_ints, sizeof(int) * (size_t)world * n, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return CORB_OK;
    }
    // device buffers; messages between a pair of ranks match in posting order; returns when this rank's sends and receives are complete
    int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs) {
        if (hub) {
            { std::lock_guard<std::mutex> lk(hub->mu); hub->posted[rank] = sends; }
            hub->barrier();
            std::vector<size_t> next(world, 0);                          // per sender: the next of its messages addressed to this rank
            int rc = CORB_OK;
            for (const Msg& r : recvs) {
                const std::vector<Msg>& from = hub->posted[r.peer];
                size_t& k = next[r.peer];
                while (k < from.size() && from[k].peer != rank) k++;
                if (k >= from.size() || from[k].bytes != r.bytes) { corb_set_error("corb_comm (in-process): receive from rank %d has no matching send", r.peer); rc = CORB_ERR_ARG; break; }
                if (r.bytes && hipMemcpyAsync(r.ptr, from[k].ptr, r.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) { corb_set_error("corb_comm (in-process): device-to-device copy failed"); rc = CORB_ERR_HIP; break; }
                k++;
            }
            if (hipStreamSynchronize(stream) != hipSuccess && rc == CORB_OK) { corb_set_error("corb_comm (in-process): stream synchronisation failed"); rc = CORB_ERR_HIP; }
            hub->barrier();                                              // the senders' buffers are free again
            return rc;
        }
        NCCLCHK(rccl().GroupStart());
        for (const Msg& m : sends) NCCLCHK(rccl().Send(m.ptr, m.bytes, NCCL_INT8, m.peer, comm, stream));
        for (const Msg& m : recvs) NCCLCHK(rccl().Recv(m.ptr, m.bytes, NCCL_INT8, m.peer, comm, stream));
        NCCLCHK(rccl().GroupEnd());
        HIPCHK(hipStreamSynchronize(stream));
        return CORB_OK;
    }
};

extern "C" int corb_comm_unique_id(void* id128)
{
    if (!id128) return CORB_ERR_ARG;
    if (!rccl().ok) { corb_set_error("%s", rccl().why.c_str()); return CORB_ERR_HIP; }
    ncclUniqueId_ id; NCCLCHK(rccl().GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return CORB_OK;
}
extern "C" int corb_comm_create(const void* id128, int rank, int world, int device, CorbComm** out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { corb_set_error("corb_comm_create: bad argument"); return CORB_ERR_ARG; }
    *out = nullptr;
    if (!rccl().ok) { corb_set_error("%s", rccl().why.c_str()); return CORB_ERR_HIP; }
    int rc = corb_select_device(device); if (rc) return rc;
    CorbComm* c = new CorbComm(); c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId_ id; memcpy(id.internal, id128, 128);
    if (rccl().CommInitRank(&c->comm, world, id, rank) != 0) { corb_set_error("ncclCommInitRank failed (rank %d of %d)", rank, world); delete c; return CORB_ERR_HIP; }
    c->d_ints_cap = 16 * (world + 1);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d_ints, sizeof(int) * (size_t)c->d_ints_cap) != hipSuccess) {
        corb_set_error("corb_comm_create: stream / buffer allocation failed"); (void)rccl().CommDestroy(c->comm); delete c; return CORB_ERR_HIP;
    }
    *out = c;
    return CORB_OK;
}
extern "C" int corb_comm_create_local(int world, const int* devices, CorbComm** out)
{
    if (!out || world < 1 || world > 1024) { corb_set_error("corb_comm_create_local: bad argument"); return CORB_ERR_ARG; }
    for (int r = 0; r < world; r++) out[r] = nullptr;
    auto hub = std::make_shared<LocalHub>(); hub->world = world; hub->posted.resize(world);
    for (int r = 0; r < world; r++) {
        const int dev = devices ? devices[r] : 0;
        int rc = corb_select_device(dev);
        CorbComm* c = nullptr;
        if (rc == CORB_OK) {
            c = new CorbComm(); c->rank = r; c->world = world; c->device = dev; c->hub = hub;
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { corb_set_error("corb_comm_create_local: stream creation failed"); delete c; c = nullptr; rc = CORB_ERR_HIP; }
        }
        if (!c) { for (int q = 0; q < r; q++) { corb_comm_destroy(out[q]); out[q] = nullptr; } return rc; }
        out[r] = c;
    }
    return CORB_OK;
}
extern "C" void corb_comm_destroy(CorbComm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_ints) (void)hipFree(c->d_ints);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}
extern "C" int corb_comm_rank(const CorbComm* c) { return c ? c->rank : -1; }
extern "C" int corb_comm_world(const CorbComm* c) { return c ? c->world : 0; }

// ---- the push's bookkeeping: pure arithmetic on what the ranks contributed to the header all-gather ----
extern "C" int corb_map_push_plan(int world, int root, const CorbPushHeader* h, int kf_capacity, int mp_capacity, const int32_t* kf_dst_first, const int32_t* mp_dst_first,
                                  int* failing_rank)
{
    if (failing_rank) *failing_rank = -1;
    if (world < 1 || root < 0 || root >= world || !h) { corb_set_error("corb_map_push_plan: bad argument"); return CORB_ERR_ARG; }
    auto fail = [&](int r, int code) { if (failing_rank) *failing_rank = r; return code; };
    // 1. a rank whose own arguments are bad: everybody returns its status
    for (int r = 0; r < world; r++) if (h[r].status != 0) { corb_set_error("map push: rank %d rejected its arguments (status %d)", r, h[r].status); return fail(r, h[r].status); }
    // 2. counts and record sizes (a record is one memcpy: both ends must agree on its size)
    bool any_mp = false;
    for (int r = 0; r < world; r++) {
        if (h[r].n_kf < 0 || h[r].n_mp < 0) { corb_set_error("map push: rank %d announces a negative count", r); return fail(r, CORB_ERR_ARG); }
        if (h[r].n_kf > 0 && h[r].kf_record_bytes != h[root].kf_record_bytes) {
            corb_set_error("map push: rank %d sends keyframe records of %d bytes, the root's store holds records of %d bytes (max_features differ)", r, h[r].kf_record_bytes, h[root].kf_record_bytes);
            return fail(r, CORB_ERR_ARG);
        }
        if (h[r].n_mp > 0 && h[r].mp_record_bytes != h[root].mp_record_bytes) {
            corb_set_error("map push: rank %d sends map-point records of %d bytes, the root's store holds records of %d bytes", r, h[r].mp_record_bytes, h[root].mp_record_bytes);
            return fail(r, CORB_ERR_ARG);
        }
        any_mp = any_mp || h[r].n_mp > 0;
    }
    // 3. placement on the root: inside its stores, ranges of different ranks disjoint
    if (!kf_dst_first || (any_mp && !mp_dst_first)) { corb_set_error("map push: the root has no destination table"); return fail(root, CORB_ERR_ARG); }
    for (int pass = 0; pass < 2; pass++) {
        const int32_t* first = pass == 0 ? kf_dst_first : mp_dst_first; const int cap = pass == 0 ? kf_capacity : mp_capacity;
        if (pass == 1 && !any_mp) break;
        for (int r = 0; r < world; r++) {
            const int n = pass == 0 ? h[r].n_kf : h[r].n_mp;
            if (n == 0) continue;
            if (first[r] < 0 || (long long)first[r] + n > cap) {
                corb_set_error("map push: rank %d sends %d %s, no room at slot %d of the root's store (capacity %d)", r, n, pass == 0 ? "keyframes" : "map points", first[r], cap);
                return fail(r, CORB_ERR_CAPACITY);
            }
            for (int q = 0; q < r; q++) {
                const int m = pass == 0 ? h[q].n_kf : h[q].n_mp;
                if (m > 0 && first[q] < first[r] + n && first[r] < first[q] + m) {
                    corb_set_error("map push: the destination ranges of ranks %d and %d overlap", q, r);
                    return fail(r, CORB_ERR_ARG);
                }
            }
        }
    }
    return CORB_OK;
}

namespace {
// the records `slots` of a store as ONE message: a contiguous ascending run is sent in place (not on the root, whose destination ranges may cover it),
// anything else is packed into a staging buffer first
struct Outgoing { char* ptr = nullptr; char* staged = nullptr; ~Outgoing() { if (staged) (void)hipFree(staged); } };
int stage_records(const char* base, size_t rec_bytes, const int32_t* slots, int n, bool in_place_ok, hipStream_t stream, Outgoing& out)
{
    if (n <= 0) return CORB_OK;
    bool run = in_place_ok;
    for (int i = 1; i < n && run; i++) run = slots[i] == slots[0] + i;
    if (run) { out.ptr = const_cast<char*>(base) + (size_t)slots[0] * rec_bytes; return CORB_OK; }
    int* dslots = nullptr;
    HIPCHK(hipMalloc((void**)&out.staged, rec_bytes * (size_t)n + 256 + sizeof(int) * (size_t)n));
    dslots = reinterpret_cast<int*>(out.staged + ((rec_bytes * (size_t)n + 255) & ~(size_t)255));
    HIPCHK(hipMemcpyAsync(dslots, slots, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream));
    corb_launch_gather_records(base, rec_bytes, dslots, n, out.staged, stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    out.ptr = out.staged;
    return CORB_OK;
}
}

extern "C" int corb_map_push_ex(CorbComm* c, const CorbMapPush* p, int root)
{
    if (!c) { corb_set_error("corb_map_push: NULL communicator"); return CORB_ERR_ARG; }
    if (root < 0 || root >= c->world) { corb_set_error("corb_map_push: bad root"); return CORB_ERR_ARG; }      // (the same value on every rank, or the job is broken anyway)
    // ---- 1. local verdict: carried into the collective instead of returned, so that no peer waits for a rank that has left ----
    CorbPushHeader mine{0, 0, 0, 0, 0};
    const bool is_root = c->rank == root;
    std::string local_why;
    auto reject = [&](int code, const char* why) { if (mine.status == 0) { mine.status = code; local_why = why; } };
    if (!p) reject(CORB_ERR_ARG, "NULL push description");
    else {
        if (p->n_kf < 0 || p->n_mp < 0) reject(CORB_ERR_ARG, "negative count");
        if (p->n_kf > 0 && (!p->kf || !p->kf_slots)) reject(CORB_ERR_ARG, "keyframes announced without store / slots");
        if (p->n_mp > 0 && (!p->mp || !p->mp_slots)) reject(CORB_ERR_ARG, "map points announced without store / slots");
        if (is_root && !p->kf) reject(CORB_ERR_ARG, "the root needs a keyframe store");
        if (p->kf && p->kf->device != c->device) reject(CORB_ERR_ARG, "keyframe store and communicator live on different devices");
        if (p->mp && p->mp->device != c->device) reject(CORB_ERR_ARG, "map-point store and communicator live on different devices");
        if (is_root && !p->kf_dst_first) reject(CORB_ERR_ARG, "the root needs kf_dst_first[world]");
        if (mine.status == 0) {
            for (int i = 0; i < p->n_kf; i++) if (p->kf_slots[i] < 0 || p->kf_slots[i] >= p->kf->capacity) { reject(CORB_ERR_ARG, "keyframe slot out of range"); break; }
            for (int i = 0; i < p->n_mp; i++) if (p->mp_slots[i] < 0 || p->mp_slots[i] >= p->mp->capacity) { reject(CORB_ERR_ARG, "map-point slot out of range"); break; }
        }
        if (mine.status == 0) {
            mine.n_kf = p->n_kf; mine.n_mp = p->n_mp;
            mine.kf_record_bytes = p->kf ? (int)p->kf->L.bytes : 0; mine.mp_record_bytes = p->mp ? (int)p->mp->L.bytes : 0;
        }
    }
    if (corb_select_device(c->device) != CORB_OK) reject(CORB_ERR_HIP, "device selection failed");
    // ---- 2. headers of all ranks ----
    const int W = c->world;
    std::vector<CorbPushHeader> hdr(W);
    int rc = c->all_gather(reinterpret_cast<const int*>(&mine), 5, reinterpret_cast<int*>(hdr.data()));
    if (rc) return rc;                                     // the transport itself failed: nothing sensible is left to agree on
    // ---- 3. the root's verdict, adopted by everybody ----
    int verdict[2] = {CORB_OK, -1};
    std::string root_why;
    if (is_root) {
        // a root with bad arguments has already put its status into its header: the plan reports it like any other rank's
        verdict[0] = corb_map_push_plan(W, root, hdr.data(), (p && p->kf) ? p->kf->capacity : 0, (p && p->mp) ? p->mp->capacity : 0,
                                        p ? p->kf_dst_first : nullptr, p ? p->mp_dst_first : nullptr, &verdict[1]);
        if (verdict[0] == CORB_OK && p && !p->mp) for (int r = 0; r < W; r++) if (hdr[r].n_mp > 0) { verdict[0] = CORB_ERR_ARG; verdict[1] = r; corb_set_error("map push: rank %d sends map points, the root has no map-point store", r); break; }
    }
    std::vector<int> verdicts(2 * (size_t)W);
    rc = c->all_gather(verdict, 2, verdicts.data());
    if (rc) return rc;
    const int v = verdicts[2 * (size_t)root], who = verdicts[2 * (size_t)root + 1];
    if (v != CORB_OK) {
        if (who == c->rank && !local_why.empty()) corb_set_error("corb_map_push: %s", local_why.c_str());
        else if (!is_root) corb_set_error("corb_map_push: rejected for every rank (code %d, about rank %d; the root's corb_last_error() has the reason)", v, who);
        return v;
    }
    // ---- 4. records: one message per rank and store ----
    std::unique_lock<std::mutex> lk_kf, lk_mp;
    if (p->kf) { lk_kf = std::unique_lock<std::mutex>(p->kf->mu); HIPCHK(hipStreamSynchronize(p->kf->stream)); }      // pending fills of the records that are about to travel
    if (p->mp) { lk_mp = std::unique_lock<std::mutex>(p->mp->mu); HIPCHK(hipStreamSynchronize(p->mp->stream)); }
    Outgoing okf, omp;
    std::vector<Msg> sends, recvs;
    // (a staging failure after the verdict would strand the peers: the exchange below is still entered, with whatever could be staged, and the error returned after it)
    int stage_rc = CORB_OK;
    if (p->n_kf > 0) { stage_rc = stage_records(p->kf->base, p->kf->L.bytes, p->kf_slots, p->n_kf, !is_root, c->stream, okf); }
    if (stage_rc == CORB_OK && p->n_mp > 0) stage_rc = stage_records(p->mp->base, p->mp->L.bytes, p->mp_slots, p->n_mp, !is_root, c->stream, omp);
    // a rank that could not stage sends from a scratch allocation of the right size instead (contents undefined) -- or, failing that, from its store: the
    // message sizes every peer expects are kept
    if (p->n_kf > 0) sends.push_back({okf.ptr ? okf.ptr : p->kf->base, (size_t)p->n_kf * p->kf->L.bytes, root});
    if (p->n_mp > 0) sends.push_back({omp.ptr ? omp.ptr : p->mp->base, (size_t)p->n_mp * p->mp->L.bytes, root});
    if (is_root)
        for (int r = 0; r < W; r++) {
            if (hdr[r].n_kf > 0) recvs.push_back({p->kf->rec(p->kf_dst_first[r]), (size_t)hdr[r].n_kf * p->kf->L.bytes, r});
            if (hdr[r].n_mp > 0) { recvs.push_back({p->mp->rec(p->mp_dst_first[r]), (size_t)hdr[r].n_mp * p->mp->L.bytes, r}); p->mp->idt_valid = false; }    // (incoming records: the id index is stale)
        }
    rc = c->exchange(sends, recvs);
    if (rc) return rc;
    if (stage_rc) return stage_rc;
    if (is_root) {
        for (int r = 0; r < W; r++) for (int i = 0; i < hdr[r].n_kf; i++) p->kf->host[p->kf_dst_first[r] + i].header_valid = false;
        for (int r = 0; r < W; r++) { if (p->kf_recv_counts) p->kf_recv_counts[r] = hdr[r].n_kf; if (p->mp_recv_counts) p->mp_recv_counts[r] = hdr[r].n_mp; }
    }
    return CORB_OK;
}

extern "C" int corb_map_push(CorbComm* c, CorbKfStore* s, const int* slots, int n_slots, int root, const int* dst_first, int* recv_counts)
{
    CorbMapPush p; memset(&p, 0, sizeof(p));
    p.kf = s; p.kf_slots = slots; p.n_kf = n_slots; p.kf_dst_first = dst_first; p.kf_recv_counts = recv_counts;
    return corb_map_push_ex(c, &p, root);
}
