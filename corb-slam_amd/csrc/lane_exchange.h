// lane_exchange.h -- cross-lane exchanges of a wavefront's reductions with less of the LDS crossbar (gfx950).
// `__shfl_xor` of a double compiles to two ds_bpermute_b32 behind an address computation: an LDS round trip per butterfly stage, on the one LDS pipe all wavefronts of a
// compute unit share.  For the constant xor masks 1, 2, 4, 8 of a butterfly the data-parallel primitives do the same move in the vector ALU:
//   xor 1, 2 : DPP quad_perm            xor 4 : DPP row_shl:4 into banks 0, 2 + row_shr:4 into banks 1, 3            xor 8 : DPP row_ror:8
// Sums are own + partner or partner + own: the same bits as the __shfl_xor forms (tools/ubench/lane_exchange.hip checks every move against __shfl_xor on the device).
//
// xor 16 / xor 32 stay on ds_bpermute by default.  v_permlane16_swap / v_permlane32_swap (swap the odd rows / upper half of one register with the even rows / lower half of
// another; -DLX_USE_SWAP) do those two stages in the vector ALU as well: bit-equal in every test, a pose optimisation of 400 observations 0.226 -> 0.220 ms, nothing for the
// CG solve once the halving stages had moved to the DPP masks.  (History: the first swap build seemed to miscompare -- a dense global BA beside another thread's tracking calls
// returned a different chi2 / lambda sequence in 3-23 % of the calls -- and the swaps were blamed.  The cause was a race in dense_chol.hip's panel kernel that the changed
// timing exposed and that the build before any of this also hit, once in 4 500 calls; with it fixed the swap build is clean in 5 600 such calls, tools/gpu_flake.sh.  The
// default stays with the instructions every GPU generation of this family has had; the 3 % on one kernel do not pay for a second code path to keep tested.)
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned lx_v2u __attribute__((ext_vector_type(2)));
#ifndef LX_USE_SWAP
#define LX_NO_SWAP 1            // see the note on v_permlane*_swap above
#endif
#define LX_LANE_ ((int)(threadIdx.x & 63))
template <int CTRL, int BANK> __device__ __forceinline__ double lx_dpp(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
// the value of lane (l ^ M), M = 1, 2, 4, 8
template <int M> __device__ __forceinline__ double lx_xor(double v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "DPP moves stay inside a row of 16 lanes");
#ifdef LX_NO_DPP
    return __shfl_xor(v, M);
#endif
    if (M == 1) return lx_dpp<0xB1, 0xF>(v, v);
    if (M == 2) return lx_dpp<0x4E, 0xF>(v, v);
    if (M == 8) return lx_dpp<0x128, 0xF>(v, v);
    return lx_dpp<0x114, 0xA>(lx_dpp<0x104, 0x5>(v, v), v);
}
// lanes without bit 5 (bit 4): a + the partner's a; lanes with it: b + the partner's b   (the transposing butterfly's step; a == b: v + partner's v on every lane)
__device__ __forceinline__ double lx_xadd32(double a, double b)
{
#ifdef LX_NO_SWAP
    { const bool h = LX_LANE_ & 32; return (h ? b : a) + __shfl_xor(h ? a : b, 32); }
#endif
    const lx_v2u r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lx_v2u r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)r1.x, (int)r0.x) + __hiloint2double((int)r1.y, (int)r0.y);
}
__device__ __forceinline__ double lx_xadd16(double a, double b)
{
#ifdef LX_NO_SWAP
    { const bool h = LX_LANE_ & 16; return (h ? b : a) + __shfl_xor(h ? a : b, 16); }
#endif
    const lx_v2u r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lx_v2u r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)r1.x, (int)r0.x) + __hiloint2double((int)r1.y, (int)r0.y);
}
// v + the value of lane (l ^ M), any power of two below 64
template <int M> __device__ __forceinline__ double lx_add_xor(double v)
{
    if (M == 32) return lx_xadd32(v, v);
    if (M == 16) return lx_xadd16(v, v);
    return v + lx_xor<(M < 16 ? M : 1)>(v);
}
// max(v, the value of lane (l ^ M))
template <int M> __device__ __forceinline__ double lx_max_xor(double v)
{
#ifdef LX_NO_SWAP
    if (M >= 16) return fmax(v, __shfl_xor(v, M));
#endif
    if (M >= 16) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const lx_v2u r0 = M == 32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const lx_v2u r1 = M == 32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return fmax(__hiloint2double((int)r1.x, (int)r0.x), __hiloint2double((int)r1.y, (int)r0.y));
    }
    return fmax(v, lx_xor<(M < 16 ? M : 1)>(v));
}
// the butterfly sum over the 64 lanes in the order 32, 16, 8, 4, 2, 1 (what `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)` computes, bit for bit)
__device__ __forceinline__ double lx_wave_sum(double v)
{
    v = lx_xadd32(v, v); v = lx_xadd16(v, v);
    v += lx_xor<8>(v); v += lx_xor<4>(v); v += lx_xor<2>(v); v += lx_xor<1>(v);
    return v;
}
__device__ __forceinline__ double lx_wave_max(double v)
{
    v = lx_max_xor<32>(v); v = lx_max_xor<16>(v); v = lx_max_xor<8>(v); v = lx_max_xor<4>(v); v = lx_max_xor<2>(v); v = lx_max_xor<1>(v);
    return v;
}

// ---- 32-bit values (a DPP move feeding an integer operation is folded into that operation by the compiler: one instruction per butterfly stage) ----
template <int CTRL, int BANK> __device__ __forceinline__ int lx_dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK, false); }
template <int M> __device__ __forceinline__ int lx_xor_i(int v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "DPP moves stay inside a row of 16 lanes");
#ifdef LX_NO_DPP
    return __shfl_xor(v, M);
#endif
    if (M == 1) return lx_dpp_i<0xB1, 0xF>(v, v);
    if (M == 2) return lx_dpp_i<0x4E, 0xF>(v, v);
    if (M == 8) return lx_dpp_i<0x128, 0xF>(v, v);
    return lx_dpp_i<0x114, 0xA>(lx_dpp_i<0x104, 0x5>(v, v), v);
}
// {own, partner} of lane l ^ 32 (l ^ 16) in unspecified order: for commutative combinations
#ifdef LX_NO_SWAP
__device__ __forceinline__ lx_v2u lx_pair32(unsigned a, unsigned b) { const bool h = LX_LANE_ & 32; lx_v2u r; r.x = h ? b : a; r.y = (unsigned)__shfl_xor((int)(h ? a : b), 32); return r; }
__device__ __forceinline__ lx_v2u lx_pair16(unsigned a, unsigned b) { const bool h = LX_LANE_ & 16; lx_v2u r; r.x = h ? b : a; r.y = (unsigned)__shfl_xor((int)(h ? a : b), 16); return r; }
#else
__device__ __forceinline__ lx_v2u lx_pair32(unsigned a, unsigned b) { return __builtin_amdgcn_permlane32_swap(a, b, false, false); }
__device__ __forceinline__ lx_v2u lx_pair16(unsigned a, unsigned b) { return __builtin_amdgcn_permlane16_swap(a, b, false, false); }
#endif
// lanes without bit 5 (bit 4): a + the partner's a; lanes with it: b + the partner's b
__device__ __forceinline__ int lx_xadd32_i(int a, int b) { const lx_v2u r = lx_pair32((unsigned)a, (unsigned)b); return (int)(r.x + r.y); }
__device__ __forceinline__ int lx_xadd16_i(int a, int b) { const lx_v2u r = lx_pair16((unsigned)a, (unsigned)b); return (int)(r.x + r.y); }
__device__ __forceinline__ int lx_wave_sum_i(int v)
{
    v = lx_xadd32_i(v, v); v = lx_xadd16_i(v, v);
    v += lx_xor_i<8>(v); v += lx_xor_i<4>(v); v += lx_xor_i<2>(v); v += lx_xor_i<1>(v);
    return v;
}
__device__ __forceinline__ int lx_wave_min_i(int v)
{
    lx_v2u r = lx_pair32((unsigned)v, (unsigned)v); v = min((int)r.x, (int)r.y);
    r = lx_pair16((unsigned)v, (unsigned)v); v = min((int)r.x, (int)r.y);
    v = min(v, lx_xor_i<8>(v)); v = min(v, lx_xor_i<4>(v)); v = min(v, lx_xor_i<2>(v)); v = min(v, lx_xor_i<1>(v));
    return v;
}
__device__ __forceinline__ int lx_wave_max_i(int v)
{
    lx_v2u r = lx_pair32((unsigned)v, (unsigned)v); v = max((int)r.x, (int)r.y);
    r = lx_pair16((unsigned)v, (unsigned)v); v = max((int)r.x, (int)r.y);
    v = max(v, lx_xor_i<8>(v)); v = max(v, lx_xor_i<4>(v)); v = max(v, lx_xor_i<2>(v)); v = max(v, lx_xor_i<1>(v));
    return v;
}
__device__ __forceinline__ unsigned lx_wave_min_u(unsigned v)
{
    lx_v2u r = lx_pair32(v, v); v = min(r.x, r.y);
    r = lx_pair16(v, v); v = min(r.x, r.y);
    v = min(v, (unsigned)lx_xor_i<8>((int)v)); v = min(v, (unsigned)lx_xor_i<4>((int)v)); v = min(v, (unsigned)lx_xor_i<2>((int)v)); v = min(v, (unsigned)lx_xor_i<1>((int)v));
    return v;
}
__device__ __forceinline__ unsigned long long lx_wave_min_u64(unsigned long long v)
{
    {
        const lx_v2u r0 = lx_pair32((unsigned)v, (unsigned)v), r1 = lx_pair32((unsigned)(v >> 32), (unsigned)(v >> 32));
        const unsigned long long x = ((unsigned long long)r1.x << 32) | r0.x, y = ((unsigned long long)r1.y << 32) | r0.y; v = x < y ? x : y;
    }
    {
        const lx_v2u r0 = lx_pair16((unsigned)v, (unsigned)v), r1 = lx_pair16((unsigned)(v >> 32), (unsigned)(v >> 32));
        const unsigned long long x = ((unsigned long long)r1.x << 32) | r0.x, y = ((unsigned long long)r1.y << 32) | r0.y; v = x < y ? x : y;
    }
#define LX_MIN64_STEP(M) do { const unsigned long long t_ = ((unsigned long long)(unsigned)lx_xor_i<M>((int)(v >> 32)) << 32) | (unsigned)lx_xor_i<M>((int)(unsigned)v); v = t_ < v ? t_ : v; } while (0)
    LX_MIN64_STEP(8); LX_MIN64_STEP(4); LX_MIN64_STEP(2); LX_MIN64_STEP(1);
#undef LX_MIN64_STEP
    return v;
}
__device__ __forceinline__ unsigned long long lx_wave_max_u64(unsigned long long v) { return ~lx_wave_min_u64(~v); }

// Inclusive prefix sum over the 64 lanes (what `for (o = 1; o < 64; o <<= 1) { t = __shfl_up(v, o); if (lane >= o) v += t; }` computes) in six DPP additions: Hillis-Steele
// inside each row of 16 lanes (row_shr:1, 2, 4, 8 with zero fill), then the row totals carried across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
__device__ __forceinline__ int lx_wave_incl_scan_i(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}
