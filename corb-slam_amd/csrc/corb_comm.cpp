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
// The point-to-point part of a push through a table of the four functions it calls (CorbRcclFns, include/corb_accel.h): the communicator passes librccl's,
// the CPU test a recording fake (corb_comm_test_rccl_exchange).  A failing Send / Recv does not leave the group open: the remaining messages are skipped,
// ncclGroupEnd is ALWAYS called (a group left open would swallow every later collective of the process), the first error is returned.
static int rccl_exchange(const CorbRcclFns& f, void* comm, void* stream, const CorbPushMsg* sends, int ns, const CorbPushMsg* recvs, int nr, char* const* send_ptr, char* const* recv_ptr)
{
    int err = f.group_start();
    if (err != 0) { corb_set_error("ncclGroupStart failed (%d)", err); return CORB_ERR_HIP; }
    for (int i = 0; i < ns && err == 0; i++) err = f.send(send_ptr[i], (size_t)sends[i].bytes, NCCL_INT8, sends[i].peer, comm, stream);
    for (int i = 0; i < nr && err == 0; i++) err = f.recv(recv_ptr[i], (size_t)recvs[i].bytes, NCCL_INT8, recvs[i].peer, comm, stream);
    const int end = f.group_end();
    if (err != 0 || end != 0) { corb_set_error("map push: ncclSend / ncclRecv / ncclGroupEnd failed (%d / %d)", err, end); return CORB_ERR_HIP; }
    return CORB_OK;
}
extern "C" int corb_comm_test_rccl_exchange(const CorbRcclFns* fns, const CorbPushMsg* sends, int n_sends, const CorbPushMsg* recvs, int n_recvs)
{
    if (!fns || n_sends < 0 || n_recvs < 0 || (n_sends && !sends) || (n_recvs && !recvs)) return CORB_ERR_ARG;
    std::vector<char*> sp(n_sends ? n_sends : 1, nullptr), rp(n_recvs ? n_recvs : 1, nullptr);       // (addresses are not dereferenced by a fake)
    for (int i = 0; i < n_sends; i++) sp[i] = reinterpret_cast<char*>((size_t)0x1000 + (size_t)sends[i].first_record);
    for (int i = 0; i < n_recvs; i++) rp[i] = reinterpret_cast<char*>((size_t)0x1000 + (size_t)recvs[i].first_record);
    return rccl_exchange(*fns, nullptr, nullptr, sends, n_sends, recvs, n_recvs, sp.data(), rp.data());
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
    // staging buffers of the outgoing records (kept between pushes: a hipMalloc / hipFree pair per push cost more than the 1.5 MB transfer it served)
    char* stage_kf = nullptr; size_t stage_kf_cap = 0; char* stage_mp = nullptr; size_t stage_mp_cap = 0;
    int* stage_slots = nullptr; size_t stage_slots_cap = 0;
    // asynchronous push: the root's layout (corb_map_push_setup) and the push in flight
    bool layout_set = false; int layout_root = -1, layout_kf_cap = 0, layout_mp_cap = 0, layout_kf_bytes = 0, layout_mp_bytes = 0;
    std::vector<int32_t> layout_kf_first, layout_mp_first;
    const CorbKfStore* layout_kf_store = nullptr; const CorbMpStore* layout_mp_store = nullptr;      // the root's stores the layout describes (begin must be handed the same)
    hipEvent_t push_done = nullptr; bool in_flight = false; int flight_rc = CORB_OK;
    CorbMapPush flight_push; int flight_root = -1; std::vector<CorbPushHeader> flight_hdr;
    int reserve(char*& buf, size_t& cap, size_t need) {
        if (need <= cap) return CORB_OK;
        if (buf) (void)hipFree(buf);
        buf = nullptr; cap = 0;
        const size_t grow = std::max(need, (size_t)1 << 20);
        if (hipMalloc((void**)&buf, grow) != hipSuccess) { corb_set_error("corb_comm: staging buffer of %zu bytes could not be allocated", grow); return CORB_ERR_HIP; }
        cap = grow; return CORB_OK;
    }

    // every rank contributes n ints; all[r * n + k] = rank r's k-th
    int all_gather(const int* mine, int n, int* all) {
        if (world == 1) { memcpy(all, mine, sizeof(int) * (size_t)n); return CORB_OK; }      // a single rank has nothing to gather (two stream round trips otherwise)
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
    // wait = false (RCCL only): the messages are enqueued on the communicator's stream, completion is the caller's event
    // dst_a / dst_b (in-process transport): the mutexes of the stores this rank RECEIVES into, held while it copies -- between the two barriers, where no other
    // rank holds a store lock (each took its source stores' only while it packed, before the first barrier): ranks that share stores cannot deadlock
    int exchange(const std::vector<Msg>& sends, const std::vector<Msg>& recvs, bool wait = true, std::mutex* dst_a = nullptr, std::mutex* dst_b = nullptr) {
        if (hub) {
            { std::lock_guard<std::mutex> lk(hub->mu); hub->posted[rank] = sends; }
            hub->barrier();
            std::unique_lock<std::mutex> la, lb;
            if (!recvs.empty()) { if (dst_a) la = std::unique_lock<std::mutex>(*dst_a); if (dst_b && dst_b != dst_a) lb = std::unique_lock<std::mutex>(*dst_b); }
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
            if (lb.owns_lock()) lb.unlock();
            if (la.owns_lock()) la.unlock();
            hub->barrier();                                              // the senders' buffers are free again
            return rc;
        }
        std::vector<CorbPushMsg> sm(sends.size()), rm(recvs.size()); std::vector<char*> sp(sends.size() + 1), rp(recvs.size() + 1);
        for (size_t i = 0; i < sends.size(); i++) { sm[i].peer = sends[i].peer; sm[i].kind = 0; sm[i].first_record = 0; sm[i].n_records = 0; sm[i].bytes = (int64_t)sends[i].bytes; sp[i] = (char*)sends[i].ptr; }
        for (size_t i = 0; i < recvs.size(); i++) { rm[i].peer = recvs[i].peer; rm[i].kind = 0; rm[i].first_record = 0; rm[i].n_records = 0; rm[i].bytes = (int64_t)recvs[i].bytes; rp[i] = (char*)recvs[i].ptr; }
        CorbRcclFns f;
        f.group_start = rccl().GroupStart; f.group_end = rccl().GroupEnd;
        f.send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, void*)>(rccl().Send);
        f.recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, void*)>(rccl().Recv);
        const int rc = rccl_exchange(f, comm, stream, sm.data(), (int)sm.size(), rm.data(), (int)rm.size(), sp.data(), rp.data());
        if (rc) return rc;
        if (wait) HIPCHK(hipStreamSynchronize(stream));
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
    c->d_ints_cap = (2 * world + 16) * (world + 1);          // headers (5 ints per rank) and the layout of corb_map_push_setup (2 world + 6 ints per rank)
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
    if (c->stage_kf) (void)hipFree(c->stage_kf);
    if (c->stage_mp) (void)hipFree(c->stage_mp);
    if (c->stage_slots) (void)hipFree(c->stage_slots);
    if (c->push_done) (void)hipEventDestroy(c->push_done);
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
// the records `slots` of a store as ONE message: a contiguous ascending run is sent in place (not on the root, whose destination ranges may cover it), anything
// else is packed into the communicator's staging buffer (kept between pushes).  On failure *ptr is still a valid device address of the message's size whenever
// one could be had (the staging buffer, else the store itself): after the verdict the exchange MUST be entered with the announced sizes.
int stage_records(CorbComm* c, bool kf_store, const char* base, size_t rec_bytes, const int32_t* slots, int n, bool in_place_ok, hipStream_t stream, bool wait, char** ptr)
{
    *ptr = const_cast<char*>(base);
    if (n <= 0) return CORB_OK;
    bool run = in_place_ok;
    for (int i = 1; i < n && run; i++) run = slots[i] == slots[0] + i;
    if (run) { *ptr = const_cast<char*>(base) + (size_t)slots[0] * rec_bytes; return CORB_OK; }
    char*& buf = kf_store ? c->stage_kf : c->stage_mp; size_t& cap = kf_store ? c->stage_kf_cap : c->stage_mp_cap;
    int rc = c->reserve(buf, cap, rec_bytes * (size_t)n);
    if (rc) return rc;
    *ptr = buf;
    char* sl = reinterpret_cast<char*>(c->stage_slots); size_t slcap = c->stage_slots_cap * sizeof(int);
    // (keyframe and map-point slot lists share the buffer: the second list goes behind the first)
    const size_t off = kf_store ? 0 : (c->stage_slots_cap / 2) * sizeof(int);
    if (slcap < 2 * sizeof(int) * (size_t)n + off) {
        rc = c->reserve(sl, slcap, 4 * sizeof(int) * (size_t)std::max(n, 4096)); c->stage_slots = reinterpret_cast<int*>(sl); c->stage_slots_cap = slcap / sizeof(int);
        if (rc) return rc;
    }
    int* dslots = c->stage_slots + (kf_store ? 0 : c->stage_slots_cap / 2);
    if (hipMemcpyAsync(dslots, slots, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream) != hipSuccess) { corb_set_error("map push: slot list upload failed"); return CORB_ERR_HIP; }
    corb_launch_gather_records(base, rec_bytes, dslots, n, buf, stream);
    if (hipGetLastError() != hipSuccess) { corb_set_error("map push: record packing failed"); return CORB_ERR_HIP; }
    if (wait && hipStreamSynchronize(stream) != hipSuccess) { corb_set_error("map push: record packing failed"); return CORB_ERR_HIP; }
    return CORB_OK;
}
}

extern "C" int corb_map_push_messages(int world, int rank, int root, const CorbPushHeader* h, const int32_t* kf_dst_first, const int32_t* mp_dst_first,
                                      CorbPushMsg* sends, int* n_sends, CorbPushMsg* recvs, int* n_recvs)
{
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !h || !sends || !n_sends || !recvs || !n_recvs) { corb_set_error("corb_map_push_messages: bad argument"); return CORB_ERR_ARG; }
    int ns = 0, nr = 0;
    if (h[rank].n_kf > 0) sends[ns++] = CorbPushMsg{root, 0, 0, h[rank].n_kf, (int64_t)h[rank].n_kf * h[rank].kf_record_bytes};
    if (h[rank].n_mp > 0) sends[ns++] = CorbPushMsg{root, 1, 0, h[rank].n_mp, (int64_t)h[rank].n_mp * h[rank].mp_record_bytes};
    if (rank == root)
        for (int r = 0; r < world; r++) {
            if (h[r].n_kf > 0) { if (!kf_dst_first) { corb_set_error("corb_map_push_messages: the root has no keyframe destination table"); return CORB_ERR_ARG; }
                                 recvs[nr++] = CorbPushMsg{r, 0, kf_dst_first[r], h[r].n_kf, (int64_t)h[r].n_kf * h[root].kf_record_bytes}; }
            if (h[r].n_mp > 0) { if (!mp_dst_first) { corb_set_error("corb_map_push_messages: the root has no map-point destination table"); return CORB_ERR_ARG; }
                                 recvs[nr++] = CorbPushMsg{r, 1, mp_dst_first[r], h[r].n_mp, (int64_t)h[r].n_mp * h[root].mp_record_bytes}; }
        }
    *n_sends = ns; *n_recvs = nr;
    return CORB_OK;
}

namespace {
// the record exchange of a push whose verdict is CORB_OK on every rank: pack, post, (optionally) wait.  Errors after the verdict do not return early -- the peers
// are already committed to the exchange: it is entered with buffers of the announced sizes and the first error is returned after it.
int push_exchange(CorbComm* c, const CorbMapPush* p, int root, const std::vector<CorbPushHeader>& hdr, const int32_t* kf_first, const int32_t* mp_first, bool wait)
{
    const int W = c->world; const bool is_root = c->rank == root;
    int rc_first = CORB_OK;
    auto note = [&](int rc) { if (rc != CORB_OK && rc_first == CORB_OK) rc_first = rc; };
    // The stores' locks (round 5; the advisor's finding on round 4, which took none with the in-process transport and none in the asynchronous form):
    //   RCCL       : this rank's stores are locked while its records are packed and the messages are posted -- and, blocking form, until they have arrived;
    //   in-process : ranks may share stores (a server process hosting its clients), so no lock may be held across the hub's barriers: a rank locks its stores
    //                while it PACKS (always into the staging buffer: an in-place message would have to stay locked until it is read), the receiving rank locks
    //                its destination stores while it copies, between the barriers (CorbComm::exchange).
    // An asynchronous push (begin ... wait) leaves the records in flight unlocked in between: that window is the caller's to keep (include/corb_accel.h).
    std::unique_lock<std::mutex> lk_kf, lk_mp;
    if (p->kf) lk_kf = std::unique_lock<std::mutex>(p->kf->mu);
    if (p->mp) lk_mp = std::unique_lock<std::mutex>(p->mp->mu);
    // pending fills of the records that are about to travel (the stores' own streams)
    if (p->kf && hipStreamSynchronize(p->kf->stream) != hipSuccess) { corb_set_error("map push: keyframe store stream failed"); note(CORB_ERR_HIP); }
    if (p->mp && hipStreamSynchronize(p->mp->stream) != hipSuccess) { corb_set_error("map push: map-point store stream failed"); note(CORB_ERR_HIP); }
    std::vector<CorbPushMsg> sm(2), rm(2 * (size_t)W); int ns = 0, nr = 0;
    note(corb_map_push_messages(W, c->rank, root, hdr.data(), kf_first, mp_first, sm.data(), &ns, rm.data(), &nr));
    std::vector<Msg> sends, recvs;
    for (int i = 0; i < ns; i++) {
        const bool kf = sm[i].kind == 0; char* ptr = nullptr;
        note(stage_records(c, kf, kf ? p->kf->base : p->mp->base, kf ? p->kf->L.bytes : p->mp->L.bytes, kf ? p->kf_slots : p->mp_slots, sm[i].n_records, !is_root && !c->hub, c->stream, wait || (bool)c->hub, &ptr));
        sends.push_back({ptr, (size_t)sm[i].bytes, sm[i].peer});
    }
    for (int i = 0; i < nr; i++) {
        const bool kf = rm[i].kind == 0;
        recvs.push_back({kf ? p->kf->rec(rm[i].first_record) : p->mp->rec(rm[i].first_record), (size_t)rm[i].bytes, rm[i].peer});
        if (!kf) p->mp->idt_valid = false;                      // (incoming records: the id index is stale)
    }
    if (c->hub) { if (lk_mp.owns_lock()) lk_mp.unlock(); if (lk_kf.owns_lock()) lk_kf.unlock(); }       // packed (and waited for): nothing is held across the barriers
    note(c->exchange(sends, recvs, wait, p->kf ? &p->kf->mu : nullptr, p->mp ? &p->mp->mu : nullptr));
    return rc_first;
}
void push_finish_root(const CorbMapPush* p, const std::vector<CorbPushHeader>& hdr, const int32_t* kf_first, int W)
{
    { std::lock_guard<std::mutex> lk(p->kf->mu);                // (the host mirror of the headers belongs to the store's lock)
      for (int r = 0; r < W; r++) for (int i = 0; i < hdr[r].n_kf; i++) p->kf->host[kf_first[r] + i].header_valid = false; }
    for (int r = 0; r < W; r++) { if (p->kf_recv_counts) p->kf_recv_counts[r] = hdr[r].n_kf; if (p->mp_recv_counts) p->mp_recv_counts[r] = hdr[r].n_mp; }
}
// what a rank can check about its own arguments before any collective: carried into the header instead of returned, so that no peer waits for a rank that has left
void push_local_header(CorbComm* c, const CorbMapPush* p, int root, CorbPushHeader& mine, std::string& why)
{
    mine = CorbPushHeader{0, 0, 0, 0, 0};
    const bool is_root = c->rank == root;
    auto reject = [&](int code, const char* w) { if (mine.status == 0) { mine.status = code; why = w; } };
    if (!p) { reject(CORB_ERR_ARG, "NULL push description"); return; }
    if (p->n_kf < 0 || p->n_mp < 0) reject(CORB_ERR_ARG, "negative count");
    if (p->n_kf > 0 && (!p->kf || !p->kf_slots)) reject(CORB_ERR_ARG, "keyframes announced without store / slots");
    if (p->n_mp > 0 && (!p->mp || !p->mp_slots)) reject(CORB_ERR_ARG, "map points announced without store / slots");
    if (is_root && !p->kf) reject(CORB_ERR_ARG, "the root needs a keyframe store");
    if (p->kf && p->kf->device != c->device) reject(CORB_ERR_ARG, "keyframe store and communicator live on different devices");
    if (p->mp && p->mp->device != c->device) reject(CORB_ERR_ARG, "map-point store and communicator live on different devices");
    if (mine.status == 0) {
        for (int i = 0; i < p->n_kf; i++) if (p->kf_slots[i] < 0 || p->kf_slots[i] >= p->kf->capacity) { reject(CORB_ERR_ARG, "keyframe slot out of range"); break; }
        for (int i = 0; i < p->n_mp; i++) if (p->mp_slots[i] < 0 || p->mp_slots[i] >= p->mp->capacity) { reject(CORB_ERR_ARG, "map-point slot out of range"); break; }
    }
    if (mine.status == 0) {
        mine.n_kf = p->n_kf; mine.n_mp = p->n_mp;
        mine.kf_record_bytes = p->kf ? (int)p->kf->L.bytes : 0; mine.mp_record_bytes = p->mp ? (int)p->mp->L.bytes : 0;
    }
    if (corb_select_device(c->device) != CORB_OK) reject(CORB_ERR_HIP, "device selection failed");
}
}

extern "C" int corb_map_push_ex(CorbComm* c, const CorbMapPush* p, int root)
{
    if (!c) { corb_set_error("corb_map_push: NULL communicator"); return CORB_ERR_ARG; }
    if (root < 0 || root >= c->world) { corb_set_error("corb_map_push: bad root"); return CORB_ERR_ARG; }      // (the same value on every rank, or the job is broken anyway)
    // ---- 1. local verdict: carried into the collective instead of returned ----
    CorbPushHeader mine; std::string local_why;
    const bool is_root = c->rank == root;
    push_local_header(c, p, root, mine, local_why);
    if (c->in_flight && mine.status == 0) { mine = CorbPushHeader{CORB_ERR_ARG, 0, 0, 0, 0}; local_why = "an asynchronous push is in flight on this communicator (corb_map_push_wait first)"; }
    if (mine.status == 0 && is_root && !p->kf_dst_first) { mine = CorbPushHeader{CORB_ERR_ARG, 0, 0, 0, 0}; local_why = "the root needs kf_dst_first[world]"; }
    // ---- 2. headers of all ranks ----
    const int W = c->world;
    std::vector<CorbPushHeader> hdr(W);
    int rc = c->all_gather(reinterpret_cast<const int*>(&mine), 5, reinterpret_cast<int*>(hdr.data()));
    if (rc) return rc;                                     // the transport itself failed: nothing sensible is left to agree on
    // ---- 3. the root's verdict, adopted by everybody ----
    int verdict[2] = {CORB_OK, -1};
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
    // ---- 4. records: one message per rank and store (the stores' locks: push_exchange) ----
    rc = push_exchange(c, p, root, hdr, p->kf_dst_first, p->mp_dst_first, true);
    if (rc) return rc;
    if (is_root) push_finish_root(p, hdr, p->kf_dst_first, W);
    return CORB_OK;
}

// ---- asynchronous form ----
extern "C" int corb_map_push_setup(CorbComm* c, int root, CorbKfStore* kf, CorbMpStore* mp, const int32_t* kf_dst_first, const int32_t* mp_dst_first)
{
    if (!c || root < 0 || root >= c->world) { corb_set_error("corb_map_push_setup: bad argument"); return CORB_ERR_ARG; }
    const int W = c->world, n = 2 * W + 6;
    std::vector<int> mine(n, 0), all((size_t)n * W);
    if (c->rank == root) {
        if (!kf || !kf_dst_first) mine[0] = CORB_ERR_ARG;
        else {
            mine[1] = kf->capacity; mine[2] = (int)kf->L.bytes; mine[3] = mp ? mp->capacity : 0; mine[4] = mp ? (int)mp->L.bytes : 0; mine[5] = (mp && mp_dst_first) ? 1 : 0;
            for (int r = 0; r < W; r++) { mine[6 + r] = kf_dst_first[r]; mine[6 + W + r] = (mp && mp_dst_first) ? mp_dst_first[r] : 0; }
        }
    }
    if (corb_select_device(c->device) != CORB_OK) mine[0] = CORB_ERR_HIP;
    int rc = c->all_gather(mine.data(), n, all.data());
    if (rc) return rc;
    for (int r = 0; r < W; r++) if (all[(size_t)n * r] != 0) { corb_set_error("corb_map_push_setup: rank %d could not take part (status %d)", r, all[(size_t)n * r]); return all[(size_t)n * r]; }
    const int* L = &all[(size_t)n * root];
    c->layout_root = root; c->layout_kf_cap = L[1]; c->layout_kf_bytes = L[2]; c->layout_mp_cap = L[3]; c->layout_mp_bytes = L[4];
    c->layout_kf_first.assign(L + 6, L + 6 + W);
    if (L[5]) c->layout_mp_first.assign(L + 6 + W, L + 6 + 2 * W); else c->layout_mp_first.clear();
    if (!c->push_done && hipEventCreateWithFlags(&c->push_done, hipEventDisableTiming) != hipSuccess) { corb_set_error("corb_map_push_setup: event creation failed"); return CORB_ERR_HIP; }
    c->layout_kf_store = c->rank == root ? kf : nullptr; c->layout_mp_store = c->rank == root ? mp : nullptr;
    c->layout_set = true;
    return CORB_OK;
}
extern "C" int corb_map_push_begin(CorbComm* c, const CorbMapPush* p, int root)
{
    if (!c) { corb_set_error("corb_map_push_begin: NULL communicator"); return CORB_ERR_ARG; }
    if (root < 0 || root >= c->world) { corb_set_error("corb_map_push_begin: bad root"); return CORB_ERR_ARG; }      // (the same value on every rank, or the job is broken anyway)
    CorbPushHeader mine; std::string local_why;
    push_local_header(c, p, root, mine, local_why);
    // local verdicts travel in the header like the others: a rank that returned here would leave its peers in the all-gather
    auto reject_local = [&](const char* w) { if (mine.status == 0) { mine = CorbPushHeader{CORB_ERR_ARG, 0, 0, 0, 0}; local_why = w; } };
    if (!c->layout_set || root != c->layout_root) reject_local("corb_map_push_setup has not been called for this root");
    if (c->in_flight) reject_local("a push is in flight on this communicator (corb_map_push_wait first)");
    if (c->rank == root && p && c->layout_set && (p->kf != c->layout_kf_store || (p->mp && p->mp != c->layout_mp_store)))
        reject_local("the root's stores are not the ones corb_map_push_setup described (the layout's capacities and record sizes are theirs)");
    const bool was_in_flight = c->in_flight;
    const int W = c->world;
    std::vector<CorbPushHeader> hdr(W);
    int rc = c->all_gather(reinterpret_cast<const int*>(&mine), 5, reinterpret_cast<int*>(hdr.data()));      // the one host synchronisation of the call
    if (rc) return rc;
    // every rank holds the root's layout: the verdict needs no second round.  The root's header carries ITS record sizes; the plan compares the others' with them
    int who = -1;
    int v = corb_map_push_plan(W, root, hdr.data(), c->layout_kf_cap, c->layout_mp_cap, c->layout_kf_first.data(), c->layout_mp_first.empty() ? nullptr : c->layout_mp_first.data(), &who);
    if (v == CORB_OK) for (int r = 0; r < W; r++) {
        // (the root's own header says whether it brought a map-point store to THIS push: mp_record_bytes 0 = none)
        if (hdr[r].n_mp > 0 && hdr[root].mp_record_bytes == 0) { v = CORB_ERR_ARG; who = r; corb_set_error("map push: rank %d sends map points, the root passed no map-point store to this push", r); break; }
        if (hdr[r].n_kf > 0 && hdr[r].kf_record_bytes != c->layout_kf_bytes) { v = CORB_ERR_ARG; who = r; corb_set_error("map push: rank %d sends keyframe records of %d bytes, the root's layout says %d", r, hdr[r].kf_record_bytes, c->layout_kf_bytes); break; }
        if (hdr[r].n_mp > 0 && (c->layout_mp_first.empty() || hdr[r].mp_record_bytes != c->layout_mp_bytes)) { v = CORB_ERR_ARG; who = r; corb_set_error("map push: rank %d sends map-point records the root's layout has no room / size for", r); break; }
    }
    if (v == CORB_OK && (!c->layout_set || was_in_flight)) v = CORB_ERR_ARG;      // (unreachable: such a rank has put its status into its header)
    if (v != CORB_OK) { if (who == c->rank && !local_why.empty()) corb_set_error("corb_map_push_begin: %s", local_why.c_str()); return v; }
    // the headers' record sizes of the ROOT are the layout's (a root that sends nothing announces 0)
    hdr[root].kf_record_bytes = c->layout_kf_bytes; hdr[root].mp_record_bytes = c->layout_mp_bytes;
    rc = push_exchange(c, p, root, hdr, c->layout_kf_first.data(), c->layout_mp_first.empty() ? nullptr : c->layout_mp_first.data(), c->hub ? true : false);
    if (hipEventRecord(c->push_done, c->stream) != hipSuccess && rc == CORB_OK) { corb_set_error("corb_map_push_begin: event record failed"); rc = CORB_ERR_HIP; }
    c->in_flight = true; c->flight_rc = rc; c->flight_push = *p; c->flight_root = root; c->flight_hdr = hdr;
    return rc;
}
extern "C" int corb_map_push_wait(CorbComm* c)
{
    if (!c) { corb_set_error("corb_map_push_wait: NULL communicator"); return CORB_ERR_ARG; }
    if (!c->in_flight) return CORB_OK;
    int rc = c->flight_rc;
    if (corb_select_device(c->device) != CORB_OK && rc == CORB_OK) rc = CORB_ERR_HIP;
    if (hipEventSynchronize(c->push_done) != hipSuccess && rc == CORB_OK) { corb_set_error("corb_map_push_wait: the push's event failed"); rc = CORB_ERR_HIP; }
    c->in_flight = false;
    if (rc == CORB_OK && c->rank == c->flight_root) push_finish_root(&c->flight_push, c->flight_hdr, c->layout_kf_first.data(), c->world);
    return rc;
}

extern "C" int corb_map_push(CorbComm* c, CorbKfStore* s, const int* slots, int n_slots, int root, const int* dst_first, int* recv_counts)
{
    CorbMapPush p; memset(&p, 0, sizeof(p));
    p.kf = s; p.kf_slots = slots; p.n_kf = n_slots; p.kf_dst_first = dst_first; p.kf_recv_counts = recv_counts;
    return corb_map_push_ex(c, &p, root);
}
