// Pieces shared by the two AND-probe kernels (scan_probe.hip: one wave per span; scan_ring.hip: loader / consumer waves around LDS rings):
// tile geometry, the query's arithmetic as it sits in LDS, the bound behind `raw_min`, the merge into the query's shared top-k pool.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "device_types.hpp"
#include "kernel_common.hpp"

namespace vq {

__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr uint32_t kPT = 1u << kProbeTileShift;   // docs per tile (32768)
constexpr uint32_t kPTW = kPT / 32;             // bitmap words per tile and dense operand (1024)
constexpr uint32_t kPNV = kPTW / 256;           // 16-byte vectors per lane, tile and dense operand (4)
constexpr uint32_t kPRk = kPT >> kRankShift;    // rank directory entries per tile and dense operand (64: one per lane)
// shape words (what only the rare paths need — scoring a flush, recomputing raw_min — lives in LDS, not in registers)
constexpr uint32_t kShCts = 0, kShTs = 1, kShVmax = 4, kShSrc = 7, kShPrunable = 11, kShScores = 12;  // scores: 3 x u64

__device__ __forceinline__ void probe_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <uint32_t ND>
struct ProbeShape {  // the query's arithmetic, read out of LDS where it is needed
    float cts, ts[ND], vmax[ND];
    uint32_t src[ND + 1];  // summation position j (set_op.rs:393,415-416: others first, shortest last) -> 0 = cover, 1 + i = dense operand i
    bool prunable;         // the bound is monotone in the cover's raw score (cover term score > 0, no inf / NaN among its scores)
};
template <uint32_t ND>
__device__ __forceinline__ ProbeShape<ND> probe_shape(const uint32_t* sh) {
    ProbeShape<ND> S;
    S.cts = __uint_as_float(sh[kShCts]);
#pragma unroll
    for (uint32_t i = 0; i < ND; ++i) {
        S.ts[i] = __uint_as_float(sh[kShTs + i]);
        S.vmax[i] = __uint_as_float(sh[kShVmax + i]);
    }
#pragma unroll
    for (uint32_t j = 0; j <= ND; ++j) S.src[j] = sh[kShSrc + j];
    S.prunable = sh[kShPrunable] != 0u;
    return S;
}

// the AND's score with the cover's value `vc` and the dense operands' values `vd`, summed in the reference's order
template <uint32_t ND>
__device__ __forceinline__ float probe_sum(const ProbeShape<ND>& S, float vc, const float (&vd)[ND]) {
    float score = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j <= ND; ++j) {
        float v = vc;
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i)
            if (S.src[j] == 1u + i) v = vd[i];
        score += v;
    }
    return score;
}

// The OR's score (set_op.rs:169-186) of a doc that holds the cover and the operands in `mask` (bit i: operand i), every leaf a term slot of
// its own, slots in leaf order: per slot the largest value of its present leaves — at least 0 —, a slot counts when that is >= 1e-5, and the
// sum over the slots is multiplied by their number twice.
template <uint32_t ND>
__device__ __forceinline__ float probe_or_sum(const ProbeShape<ND>& S, float vc, const float (&vd)[ND], uint32_t mask) {
    float sum = 0.0f, nd = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j <= ND; ++j) {
        float v = vc;
        bool present = true;
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i)
            if (S.src[j] == 1u + i) {
                v = vd[i];
                present = ((mask >> i) & 1u) != 0u;
            }
        const float m = present ? fmaxf(0.0f, v) : 0.0f;
        if (m >= 0.00001f) nd += 1.0f;
        sum += m;
    }
    return sum * nd * nd;
}

// Smallest raw f16 score of a cover posting whose hit can still reach the threshold score `thr_f` (wave-wide 64-ary search over the
// finite non-negative f16 patterns; the bound is monotone in raw).  0: everything stays live.
// or_mask != ~0: the OR's bound of a doc that holds the cover and exactly the operands of that mask
template <uint32_t ND>
__device__ __forceinline__ uint32_t probe_raw_min(const uint32_t* sh, float thr_f, const uint32_t lane, const uint32_t or_mask = 0xFFFFFFFFu) {
    const ProbeShape<ND> S = probe_shape<ND>(sh);
    if (!S.prunable) return 0u;
    uint32_t lo = 0u, hi = 0x7C00u;
    while (hi > lo) {  // uniform
        const uint32_t step = (hi - lo + 63u) >> 6;
        const uint32_t p = lo + lane * step;
        bool dead = false;
        if (p < hi) {
            const float vc = posting_value_fast(S.cts, (uint16_t)p);
            dead = (or_mask == 0xFFFFFFFFu ? probe_sum<ND>(S, vc, S.vmax) : probe_or_sum<ND>(S, vc, S.vmax, or_mask)) < thr_f;
        }  // (p is a finite f16: the short division is exact, tests check all 2^16)  NaN threshold (none yet): never dead
        const uint32_t c = (uint32_t)__popcll(wballot(dead));
        if (c == 0u) hi = lo;
        else {
            const uint32_t last = lo + (c - 1u) * step;
            const uint32_t nhi = last + step;
            lo = last + 1u;
            hi = nhi < hi ? nhi : hi;
        }
    }
    return lo;
}

// Merge the span's best keys (candidate buffer, any order) into the query's pool under its (try-)lock; the pool's k-th key becomes the
// query's threshold (QHeader::gthr) and this span's.  k <= 32: the span's keys sit in lanes 0-31, the pool's in lanes 32-63, one
// bitonic sort over the wave, duplicates (a key this span merged in before) dropped.
// WG: the workgroup is this one wave and may use its barrier (k_scan_probe); otherwise the wave orders its own LDS accesses (in order per wave) by fences
template <bool WG>
__device__ void probe_pool_merge(const CandState& cs, uint32_t k, uint8_t* pool_bytes, const uint32_t lane) {
    uint32_t* const lock = reinterpret_cast<uint32_t*>(pool_bytes);
    uint32_t* const pn = lock + 1;
    unsigned long long* const pk = reinterpret_cast<unsigned long long*>(pool_bytes + sizeof(DPool));
    if (WG) __syncthreads();
    else probe_lds_fence();
    uint32_t n_own = *cs.n;
    n_own = n_own < cs.cap ? n_own : cs.cap;
    n_own = n_own < 32u ? n_own : 32u;  // (after a flush the buffer holds at most k <= 32 keys unless nothing was pruned yet)
    unsigned long long key = lane < n_own ? cs.cand[lane] : 0ull;
    // try-lock: a span that finds the pool busy leaves its keys for its next flush (a spinning wave would only delay the holder: every
    // attempt is a memory-side atomic on the same word)
    uint32_t got = 0u;
    if (lane == 0) {
        uint32_t expected = 0u;
        got = __hip_atomic_compare_exchange_strong(lock, &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
    if (!got) return;  // uniform
    const uint32_t gn = __hip_atomic_load(pn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane >= 32u && lane - 32u < gn) key = __hip_atomic_load(pk + (lane - 32u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // bitonic sort, descending over the lanes
#pragma unroll
    for (uint32_t size = 2; size <= 64u; size <<= 1) {
#pragma unroll
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned long long other = shfl_u64(key, lane ^ stride);
            const bool desc = (lane & size) == 0u || size == 64u;
            const bool lower = (lane & stride) == 0u;
            const unsigned long long mx = key > other ? key : other, mn = key > other ? other : key;
            key = (lower == desc) ? mx : mn;
        }
    }
    const unsigned long long prev = shfl_u64(key, (lane + 63u) & 63u);
    const bool uniq = key != 0ull && (lane == 0u || key != prev);
    const unsigned long long um = wballot(uniq);
    const uint32_t pos = (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
    const uint32_t total = (uint32_t)__popcll(um);
    if (uniq && pos < k) __hip_atomic_store(pk + pos, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long kth = 0ull;
    if (total >= k) {  // uniform: the k-th distinct key
        const unsigned long long km = wballot(uniq && pos == k - 1u);
        kth = shfl_u64(key, (uint32_t)__ffsll((long long)km) - 1u);
    }
    if (lane == 0) {
        __hip_atomic_store(pn, total < k ? total : k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kth) atomicMax(cs.gthr, kth);
    }
    if (WG) __syncthreads();
    else probe_lds_fence();
    if (lane == 0) {
        __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (kth > *cs.thr) *cs.thr = kth;
    }
    if (WG) __syncthreads();
    else probe_lds_fence();
}


}  // namespace vq
