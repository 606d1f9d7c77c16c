// store_host.h -- host-side objects of the device-resident stores (corb_store.cpp, corb_comm.cpp, corb_ba_store.cpp)
#pragma once
#include "store_internal.h"
#include <mutex>
#include <vector>
#include "device_util.h"

struct CorbKfStore {
    int device = 0, capacity = 0, F = 0;
    RecLayout L{1};
    char* base = nullptr;                     // [capacity][L.bytes]
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    std::mutex mu;
    struct Host { int n = -1, n_nodes = 0; unsigned long long id = 0; std::vector<uint32_t> node_id; bool header_valid = false; };
    std::vector<Host> host;                   // host mirror of the small parts (counts, vocabulary node ids)
    char* rec(int slot) const { return base + (size_t)slot * L.bytes; }
};


struct CorbMpStore {
    int device = 0, capacity = 0, O = 0;      // O = max observations per map point
    MpLayout L{1};
    char* base = nullptr;                     // [capacity][L.bytes]
    hipStream_t stream = nullptr;
    std::mutex mu;
    CorbIdTable idt{nullptr, nullptr, 0};     // mnId -> slot of the slots indexed by corb_mp_store_build_index (tracking calls on records); keys == nullptr: none
    int idt_first = 0, idt_n = 0; bool idt_valid = false;      // cleared by whatever rewrites record headers (put, incoming push)
    // scratch of corb_local_ba_store (a call per keyframe: no hipMalloc / hipFree per call): device arena + page-locked staging, grown between calls to
    // what the largest call asked for (up to 256 MB / 64 MB; beyond that the call allocates)
    char* lba_dev = nullptr; size_t lba_dev_cap = 0, lba_dev_want = 0;
    char* lba_host = nullptr; size_t lba_host_cap = 0, lba_host_want = 0;
    hipEvent_t lba_event = nullptr;                // the window's graph is complete (corb_local_ba_store: this store's stream -> the optimiser's)
    char* rec(int slot) const { return base + (size_t)slot * L.bytes; }
};
