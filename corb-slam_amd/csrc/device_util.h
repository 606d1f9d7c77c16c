// device_util.h -- small device-wide primitives shared by the host code (exclusive scan, open-addressing id table)
#pragma once
#include "corb_internal.h"

// out[i] = sum of in[0..i) for i in [0, n]; out has n + 1 entries (out[n] = total).  in == out is allowed (in-place on the first n entries).
// scratch: at least corb_scan_scratch_ints(n) ints.  Three launches; sums are 32-bit (the caller checks the total against its own bound).
size_t corb_scan_scratch_ints(size_t n);
void corb_launch_exclusive_scan(const int* in, int* out, size_t n, int* scratch, hipStream_t s);
// up to four independent scans of at most 32 768 counts each in one launch (false: too long -- nothing was launched, scan them one by one)
bool corb_launch_exclusive_scan4(const int* const* in, int* const* out, const size_t* n, int count, hipStream_t s);

// 64-bit id -> index table in device memory (open addressing, linear probing).  cap = a power of two >= 2 x entries; keys[] initialised to CORB_IDTAB_EMPTY.
#define CORB_IDTAB_EMPTY 0xFFFFFFFFFFFFFFFFull
struct CorbIdTable { unsigned long long* keys; int* vals; unsigned int mask; };
__device__ __forceinline__ unsigned int corb_idtab_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}
// returns the value stored for the key, or -1
__device__ __forceinline__ int corb_idtab_find(const CorbIdTable& t, unsigned long long key)
{
    unsigned int h = corb_idtab_hash(key) & t.mask;
    for (;;) {
        const unsigned long long k = t.keys[h];
        if (k == key) return t.vals[h];
        if (k == CORB_IDTAB_EMPTY) return -1;
        h = (h + 1) & t.mask;
    }
}
// inserts (key, val) into a table whose vals were preset to 0x7F7F7F7F: of several writers of one key the LOWEST val stays (deterministic, unlike corb_idtab_insert)
__device__ __forceinline__ void corb_idtab_insert_min(const CorbIdTable& t, unsigned long long key, int val)
{
    unsigned int h = corb_idtab_hash(key) & t.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&t.keys[h], CORB_IDTAB_EMPTY, key);
        if (prev == CORB_IDTAB_EMPTY || prev == key) { atomicMin(&t.vals[h], val); return; }
        h = (h + 1) & t.mask;
    }
}
// inserts (key, val); returns false if the key is already present (the first writer's value stays)
__device__ __forceinline__ bool corb_idtab_insert(const CorbIdTable& t, unsigned long long key, int val)
{
    unsigned int h = corb_idtab_hash(key) & t.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&t.keys[h], CORB_IDTAB_EMPTY, key);
        if (prev == CORB_IDTAB_EMPTY) { t.vals[h] = val; return true; }
        if (prev == key) return false;
        h = (h + 1) & t.mask;
    }
}
