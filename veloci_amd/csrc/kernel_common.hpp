// Device helpers shared by the kernel translation units (kernels.hip, scan_probe.hip): address-space casts, wave scans,
// the LDS candidate buffer with its bitonic prune, the posting-value arithmetic.  Everything is static / inline: every
// translation unit gets its own copy (no relocatable device code).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "device_types.hpp"

namespace vq {
// Pointers read out of the query blob are generic ("flat") to the compiler; every one of them points into
// HBM.  Casting to the global address space turns flat_load (which also ties up the LDS counter) into
// global_load.
#define VQ_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ const VQ_GLOBAL T* as_global(const T* p) {
    return (const VQ_GLOBAL T*)p;
}
// Uniform (same for the whole wave) read-only descriptors are read through the constant address space: with a wave-uniform address
// these are scalar loads (SGPR results, scalar cache) instead of one LDS / vector access per lane.
#define VQ_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const VQ_CONST T* as_const(const void* p) {
    return (const VQ_CONST T*)(uintptr_t)p;
}
// ------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// First index in a[0..n) with a[idx] >= target (a ascending).  Whole wave cooperates: 64 probes per
// round, ~log64(n) dependent rounds instead of log2(n).
__attribute__((unused)) static __device__ uint32_t wave_lower_bound(const uint32_t* __restrict__ a_, uint32_t n, uint32_t target) {
    const VQ_GLOBAL uint32_t* a = as_global(a_);
    uint32_t lo = 0, hi = n;
    const uint32_t lane = lane_id();
    while (hi > lo) {
        uint32_t range = hi - lo;
        uint32_t step = (range + 63u) >> 6;
        uint32_t p = lo + lane * step;
        bool less = false;
        if (p < hi) less = a[p] < target;
        unsigned long long m = __ballot(less);
        uint32_t c = (uint32_t)__popcll(m);
        if (c == 0) {
            hi = lo;
        } else {
            uint32_t last = lo + (c - 1u) * step;
            uint32_t nhi = last + step;
            lo = last + 1u;
            hi = nhi < hi ? nhi : hi;
        }
    }
    return lo;
}
// ------------------------------------------------------------------------------------ candidate buffer
// LDS buffer of 64-bit keys; prune = bitonic sort (descending) + keep k + raise the threshold.
struct CandState {
    unsigned long long* cand;  // [cap]
    uint32_t* n;               // pushes so far (may exceed cap)
    unsigned long long* thr;   // keys <= thr cannot enter the top-k any more
    uint32_t cap;              // power of two, >= 2 * k
    unsigned long long* gthr = nullptr;  // the query's threshold word in HBM, shared by all of its spans (QHeader::gthr)
    unsigned long long upper = ~0ull;    // keys at or above this never enter (QHeader::key_upper: pages of a deep request)
};

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, uint32_t src_lane) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src_lane), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// All threads of the workgroup call this together.  Without `force` a buffer that already holds <= k keys
// is left as it is (the caller only needs the SET of the best k); with `force` the keys end up sorted
// descending.  On return *cs.n <= k.
__attribute__((unused)) static __device__ void cand_prune(const CandState& cs, uint32_t k, bool force = false) {
    __syncthreads();
    uint32_t n = *cs.n;
    if (n > cs.cap) n = cs.cap;
    __syncthreads();
    if (n <= k && !force) {
        if (threadIdx.x == 0) *cs.n = n;
        __syncthreads();
        return;
    }
    uint32_t m = 2;
    while (m < n) m <<= 1;
    for (uint32_t i = n + threadIdx.x; i < m; i += kBlock) cs.cand[i] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += kBlock) {
                uint32_t i = ((t & ~(stride - 1u)) << 1) | (t & (stride - 1u));  // (stride is a power of two)
                uint32_t j = i + stride;
                bool desc = (i & size) == 0;
                unsigned long long a = cs.cand[i], b = cs.cand[j];
                if ((a < b) == desc) {
                    cs.cand[i] = b;
                    cs.cand[j] = a;
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        *cs.n = n < k ? n : k;
        if (n >= k && k > 0) {
            unsigned long long t = cs.cand[k - 1];
            if (cs.gthr) {  // publish, and adopt what another span of the query has reached
                const unsigned long long other = atomicMax(cs.gthr, t);
                t = other > t ? other : t;
            }
            *cs.thr = t;
        }
    }
    __syncthreads();
}
// ------------------------------------------------------------------------------------ score arithmetic
__device__ __forceinline__ float posting_value(float term_score, uint16_t f16bits) {
    // search_field.rs:426  hit.score * (el.score.to_f32() / 100.0)
    float a = __half2float(__ushort_as_half(f16bits));
    return term_score * (a / 100.0f);
}

// a / 100.0f for a = any finite f16 value, without the generic division sequence: q0 = a * RN(1/100), one exact-remainder
// correction (Markstein).  tests/test_gpu_parity.py checks it against the correctly rounded division for ALL 2^16 f16 inputs.
__device__ __forceinline__ float div100_fast(float a) {
    const float rb = 0.01f;
    const float q0 = a * rb;
    const float r = __builtin_fmaf(-100.0f, q0, a);
    return a == 0.0f ? q0 : __builtin_fmaf(r, rb, q0);  // keeps the sign of a zero
}
__device__ __forceinline__ float posting_value_fast(float term_score, uint16_t f16bits) {
    return term_score * div100_fast(__half2float(__ushort_as_half(f16bits)));
}
// ------------------------------------------------------------------------------------ per-hit scoring
// wave64 inclusive add-scan with DPP row shifts / row broadcasts (gfx9 family), ~6 VALU instead of 6 LDS permutes
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xE, false);  // row_shr:4, banks 1-3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xC, false);  // row_shr:8, banks 2-3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1,3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2,3
    return x;
}
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t x, uint32_t* total) {
    const uint32_t incl = wave_incl_scan_u32(x);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    return incl - x;
}

// A DOp fetched as ten dwords through the constant address space (scalar loads), fields decoded with scalar shifts.
struct KOp {
    uint32_t r0, r1;
    unsigned long long slots[2], order[2];
    __device__ __forceinline__ explicit KOp(const VQ_CONST DOp* p) {
        static_assert(sizeof(DOp) == 40 && kMaxChildren == 16, "DOp layout");
        const VQ_CONST uint32_t* w = (const VQ_CONST uint32_t*)p;
        r0 = w[0];
        r1 = w[1];
        slots[0] = ((unsigned long long)w[3] << 32) | w[2];
        slots[1] = ((unsigned long long)w[5] << 32) | w[4];
        order[0] = ((unsigned long long)w[7] << 32) | w[6];
        order[1] = ((unsigned long long)w[9] << 32) | w[8];
    }
    __device__ __forceinline__ uint32_t kind() const { return r0 & 0xFFu; }
    __device__ __forceinline__ uint32_t nchild() const { return (r0 >> 8) & 0xFFu; }
    __device__ __forceinline__ uint32_t nslots() const { return (r0 >> 16) & 0xFFu; }
    __device__ __forceinline__ uint32_t list_begin() const { return r1 & 0xFFFFu; }
    __device__ __forceinline__ uint32_t list_count() const { return r1 >> 16; }
    __device__ __forceinline__ uint32_t child_slot(uint32_t k) const { return (uint32_t)((k < 8u ? slots[0] : slots[1]) >> (8u * (k & 7u))) & 0xFFu; }
    __device__ __forceinline__ uint32_t and_order(uint32_t k) const { return (uint32_t)((k < 8u ? order[0] : order[1]) >> (8u * (k & 7u))) & 0xFFu; }
};

}  // namespace vq
