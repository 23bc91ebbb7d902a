// CDNA4 (gfx950) kernels of the veloci query path.  wave64 everywhere; no MFMA (integer / gather /
// scalar-f32 work, HBM bound).  Compiled with -ffp-contract=off and correctly-rounded f32 division so
// that every score is bit-identical to the reference's scalar Rust arithmetic.
//
//   k_tile_scan    K1+K2+K3+K4+K5+K6+K7+K10+K11 fused: one workgroup owns a contiguous span of the
//                  shard's doc-id space for one query and walks it tile by tile:
//                    stream the doc ids of every list that falls into the tile (16 B/lane coalesced
//                    loads, several lists in flight per lane) -> per-list LDS bitmaps -> postfix presence
//                    program on bitmap words (AND/OR/filter) -> for surviving docs only: rank = prefix
//                    popcount -> gather the f16 anchor scores -> reference score arithmetic -> boosts ->
//                    facet histogram -> per-workgroup exact top-k (64-bit keys, LDS candidate buffer +
//                    bitonic prune)
//   k_merge_spans  per query: merge the span-local top-k lists into the shard partial
//   k_finalize     per query: merge the partials of all shards (after the RCCL all-gather) into the
//                  final ranked hits; sum hit counts
//   k_facet_select per (query, facet): top-`top` histogram entries (count desc, value id asc)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "device_types.hpp"
#include "kernels.hpp"
#include "kernel_common.hpp"

namespace vq {


// Diagnostic build only (make STAMP=1): cycle shares of the scan phases as seen by wave 0 of every workgroup.
#ifdef VQ_STAMP
__device__ unsigned long long g_stamp[16];
#define VQ_STAMP_INIT                                       \
    unsigned long long _st0 = __builtin_amdgcn_s_memtime(); \
    unsigned long long _acc[16] = {0};
#define VQ_STAMP_AT(k)                                              \
    {                                                               \
        unsigned long long _st1 = __builtin_amdgcn_s_memtime();     \
        _acc[k] += _st1 - _st0;                                     \
        _st0 = _st1;                                                \
    }
#define VQ_STAMP_COUNT(k) _acc[k] += 1ull;
#define VQ_STAMP_FLUSH                                                  \
    if (threadIdx.x == 0) {                                             \
        _Pragma("unroll") for (int _k = 0; _k < 16; ++_k) if (_acc[_k]) atomicAdd(&g_stamp[_k], _acc[_k]); \
    }
void debug_read_stamps(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof z);
    }
}
#else
#define VQ_STAMP_INIT
#define VQ_STAMP_AT(k)
#define VQ_STAMP_COUNT(k)
#define VQ_STAMP_FLUSH
#endif






__global__ void k_div100_check(uint32_t* mismatches) {
    const uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;
    if (bits >= 65536u) return;
    const float a = __half2float(__ushort_as_half((uint16_t)bits));
    if (a != a || a - a != 0.0f) return;  // NaN / inf
    const float exact = a / 100.0f;
    const float fast = div100_fast(a);
    if (__float_as_uint(exact) != __float_as_uint(fast)) atomicAdd(mismatches, 1u);
}
uint32_t debug_div100_mismatches() {
    uint32_t* d = nullptr;
    uint32_t h = 0xFFFFFFFFu;
    if (hipMalloc(&d, 4) != hipSuccess) return h;
    (void)hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k_div100_check, dim3(256), dim3(256), 0, 0, d);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return h;
}

__device__ __forceinline__ float log10_f32(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float log2_f32(float x) { return (float)log2((double)x); }

// apply_boost (boost.rs:283-377): boost function, then the optional expression
__device__ float apply_boost_value(float score, const DColBoost& cb, float v) {
    float vp = v + cb.param;
    switch (cb.fun) {
        case BF_LOG10: score *= log10_f32(vp); break;
        case BF_LOG2: score *= log2_f32(vp); break;
        case BF_MULTIPLY: score *= vp; break;
        case BF_ADD: score += vp; break;
        case BF_REPLACE: score = vp; break;
        default: break;
    }
    if (cb.expr_op != EX_NONE) {
        float l = cb.expr_lkind == 0 ? v : cb.expr_lval;
        float r = cb.expr_rkind == 0 ? v : cb.expr_rval;
        float e;
        switch (cb.expr_op) {
            case EX_DIV: e = l / r; break;
            case EX_MUL: e = l * r; break;
            case EX_ADD: e = l + r; break;
            default: e = l - r; break;
        }
        score += e;
    }
    return score;
}

__device__ float apply_col_boost(float score, const DColBoost& cb, uint32_t doc) {
    // add_boost boost.rs:470-504
    for (uint32_t s = 0; s < cb.nskip; ++s)
        if (fabsf(cb.skip[s] - score) < 0.00001f) return score;
    if (doc < cb.key_base) return score;
    uint32_t row = doc - cb.key_base;
    if (row >= cb.num_keys) return score;
    if (cb.present && !((as_global(cb.present)[row >> 5] >> (row & 31u)) & 1u)) return score;
    return apply_boost_value(score, cb, as_global(cb.values)[row]);
}

constexpr uint32_t kFacetCache = 1024;  // slots of the per-workgroup facet counter cache (keys + counts: 8 KB)
// Facet counting (persistence.rs:164-175): a hit adds one to the histogram entry of each of its values.  Hot values (a Zipf-skewed category
// field) would serialise on single HBM words; the workgroup keeps a direct-mapped counter cache in LDS — a value takes the slot its index
// hashes to if that is free or already its own, else the add goes straight to HBM — and flushes the cache once, when its span ends.
__device__ __forceinline__ void facet_add(uint32_t* fc_keys, uint32_t* hist, uint32_t idx) {
    if (fc_keys) {  // uniform
        const uint32_t slot = (idx * 2654435761u) >> 22;
        const uint32_t old = atomicCAS(&fc_keys[slot], 0xFFFFFFFFu, idx);
        if (old == 0xFFFFFFFFu || old == idx) {
            atomicAdd(&fc_keys[kFacetCache + slot], 1u);
            return;
        }
    }
    atomicAdd(&hist[idx], 1u);
}
__device__ __forceinline__ void facet_cache_flush(uint32_t* fc_keys, uint32_t* hist) {
    for (uint32_t s2 = threadIdx.x; s2 < kFacetCache; s2 += 64u) {
        const uint32_t key = fc_keys[s2];
        if (key != 0xFFFFFFFFu) atomicAdd(&hist[key], fc_keys[kFacetCache + s2]);
    }
}


struct ScoreCtx {
    const DList* lists;
    const DOp* ops;
    uint32_t n_ops;
    uint32_t simple_n;  // != 0: ops = simple_n single-list posting leaves (+ one AND/OR root when simple_n > 1)
    const DGroup* groups;
    uint32_t n_groups;
    const DTermBoost* tboosts;
    uint32_t n_tboost;
    const DColBoost* cols;
    uint32_t n_col;
    const DLocField* locf;
    const uint16_t* loc_idx;
    uint32_t n_locf;
    const DFacet* facets;
    uint32_t n_facets;
    const uint32_t* bm;
    const uint16_t* pre;
    const uint32_t* cur;
    const uint32_t* cnt_lo;
    uint32_t WW;
    float* fstack;  // this lane's column, stride kBlock
    uint32_t* hist;
    const VQ_CONST DOp* kops;      // the same ops / lists in the query blob in HBM, for wave-uniform (scalar) reads
    const VQ_CONST DList* klists;
    uint32_t* gb;                  // this lane's count of bytes read by per-hit gathers (profiling: QHeader::stat_off)
    uint32_t* fc_keys;             // facet counter cache in LDS (null: none)
};

__device__ __forceinline__ float pick4(float v0, float v1, float v2, float v3, uint32_t k) { return k == 0 ? v0 : k == 1 ? v1 : k == 2 ? v2 : v3; }

// flat AND / OR / single leaf over <= 4 single-list posting leaves: all gathers are issued together
__device__ __forceinline__ float tree_score_simple(const ScoreCtx& c, uint32_t w, uint32_t b) {
    const uint32_t n = c.simple_n;
    const uint32_t below = (1u << b) - 1u;
    uint32_t presm = 0;
    uint32_t idx[4] = {0, 0, 0, 0};
    uint32_t li[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k < n) {
            li[k] = c.ops[k].list_begin;
            const uint32_t word = c.bm[li[k] * c.WW + w];
            if ((word >> b) & 1u) {
                presm |= 1u << k;
                idx[k] = c.cur[li[k]] + c.cnt_lo[li[k]] + (uint32_t)c.pre[li[k] * c.WW + w] + (uint32_t)__popc(word & below);
            }
        }
    }
    uint16_t raw[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if ((presm >> k) & 1u) raw[k] = as_global(c.lists[li[k]].scores)[idx[k]];
    *c.gb += 2u * (uint32_t)__popc(presm);
    float val[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) val[k] = posting_value(c.lists[li[k]].term_score, raw[k]);
    if (n == 1) return val[0];
    const DOp& root = c.ops[n];
    if (root.kind == OP_AND) {  // set_op.rs:415-416: others summed first, the shortest list's score last
        float s = 0.0f;
        for (uint32_t k = 0; k < n; ++k) s += pick4(val[0], val[1], val[2], val[3], root.and_order[k]);
        return s;
    }
    float sum = 0.0f, nd = 0.0f;  // set_op.rs:169-186
    for (uint32_t slot = 0; slot < root.nslots; ++slot) {
        float m = 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n && root.child_slot[k] == slot && ((presm >> k) & 1u)) m = fmaxf(m, val[k]);
        if (m >= 0.00001f) nd += 1.0f;
        sum += m;
    }
    return sum * nd * nd;
}

// Where a hit's list memberships and posting indices come from.
struct TileHit {  // a doc of the tile being scanned: the tile bitmaps and their prefix popcounts
    const ScoreCtx& c;
    uint32_t w, b;
    __device__ __forceinline__ bool present(uint32_t li) const { return (c.bm[li * c.WW + w] >> b) & 1u; }
    __device__ __forceinline__ uint32_t index(uint32_t li) const {
        const uint32_t word = c.bm[li * c.WW + w];
        return c.cur[li] + c.cnt_lo[li] + (uint32_t)c.pre[li * c.WW + w] + (uint32_t)__popc(word & ((1u << b) - 1u));
    }
};
constexpr uint32_t kQueueCap = 96;       // k_tile_scan: survivors wait here until 64 of them make a full scoring round (< 64 waiting + 32 new)
constexpr uint32_t kQueueMaxLists = 16;  // queries with more lists score each tile's survivors at once
struct QueuedHit {  // a doc queued by an earlier tile: memberships as a mask, posting indices captured when it was queued
    unsigned long long mask;
    const uint32_t* qidx;  // [L][kQueueCap]
    uint32_t slot;
    __device__ __forceinline__ bool present(uint32_t li) const { return (mask >> li) & 1ull; }
    __device__ __forceinline__ uint32_t index(uint32_t li) const { return qidx[li * kQueueCap + slot]; }
};

// any tree: postfix interpreter; stack slot index is uniform across lanes
template <class HitT>
__device__ float tree_score_generic(const ScoreCtx& c, const HitT& hit) {
    uint32_t sp = 0;
    uint32_t pmask = 0;  // bit s: stack slot s holds a present value
    for (uint32_t o = 0; o < c.n_ops; ++o) {
        const KOp op(c.kops + o);  // (scalar loads: o is uniform)
        float s = 0.0f;
        bool present = false;
        if (op.kind() == OP_LEAF) {
            for (uint32_t j = 0; j < op.list_count(); ++j) {
                const uint32_t li = op.list_begin() + j;
                if (hit.present(li)) {
                    float v = 0.0f;
                    const uint32_t lflags = c.klists[li].flags;
                    if (lflags & LIST_HAS_SCORES) {
                        const uint32_t idx = hit.index(li);
                        const uint16_t* sp16 = c.klists[li].scores;
                        if (lflags & LIST_F32) v = as_global(reinterpret_cast<const float*>(sp16))[idx];
                        else v = posting_value(c.klists[li].term_score, as_global(sp16)[idx]);
                        *c.gb += (lflags & LIST_F32) ? 4u : 2u;
                    }
                    if (!present || v > s) s = v;  // dedup keeps the max (search_field.rs:455-461)
                    present = true;
                }
            }
        } else if (op.kind() == OP_BOOST1N) {  // apply_boost_values_anchor (boost.rs:255-281): one boost value per anchor of the leaf below
            const uint32_t top = sp - 1u;
            if ((pmask >> top) & 1u) {
                const uint32_t li = op.list_begin();
                if (hit.present(li)) {
                    const float v = as_global(reinterpret_cast<const float*>(c.klists[li].scores))[hit.index(li)];
                    *c.gb += 4u;
                    c.fstack[top * kBlock] = apply_boost_value(c.fstack[top * kBlock], c.cols[op.child_slot(0)], v);
                }
            }
            continue;
        } else if (op.kind() == OP_AND) {
            const uint32_t base = sp - op.nchild();
            present = true;
            for (uint32_t k = 0; k < op.nchild(); ++k) present = present && ((pmask >> (base + k)) & 1u);
            if (present) {
                s = 0.0f;  // set_op.rs:415-416
                for (uint32_t k = 0; k < op.nchild(); ++k) s += c.fstack[(base + op.and_order(k)) * kBlock];
            }
            sp = base;
        } else {
            const uint32_t base = sp - op.nchild();
            float sum = 0.0f;
            float nd = 0.0f;
            for (uint32_t slot = 0; slot < op.nslots(); ++slot) {  // set_op.rs:169-186
                float m = 0.0f;
                for (uint32_t k = 0; k < op.nchild(); ++k) {
                    if (op.child_slot(k) == slot && ((pmask >> (base + k)) & 1u)) {
                        present = true;
                        m = fmaxf(m, c.fstack[(base + k) * kBlock]);
                    }
                }
                if (m >= 0.00001f) nd += 1.0f;
                sum += m;
            }
            s = sum * nd * nd;
            sp = base;
        }
        c.fstack[sp * kBlock] = s;
        pmask = present ? (pmask | (1u << sp)) : (pmask & ~(1u << sp));
        ++sp;
    }
    return c.fstack[0];
}

// sink stages in the reference's order: column boosts, phrase groups, term boosts, text locality, facets
template <class HitT>
__device__ float sink_stages(const ScoreCtx& c, float score, uint32_t doc, const HitT& hit) {
    for (uint32_t k = 0; k < c.n_col; ++k) score = apply_col_boost(score, c.cols[k], doc);
    *c.gb += 4u * c.n_col;
    for (uint32_t g = 0; g < c.n_groups; ++g) {
        bool in = false;
        for (uint32_t j = 0; j < c.groups[g].list_count; ++j) in = in || hit.present(c.groups[g].list_begin + j);
        if (in) score *= c.groups[g].mult;
    }
    for (uint32_t t = 0; t < c.n_tboost; ++t)
        if (hit.present(c.tboosts[t].list)) score *= c.tboosts[t].mult;
    if (c.n_locf) {  // boost.rs:11-87: 2*c*c per field with c > 1, the MINIMUM over fields (:25)
        float best = 0.0f;
        bool have = false;
        for (uint32_t f = 0; f < c.n_locf; ++f) {
            if (c.locf[f].list_count == kLocPrecomputed) {  // field whose text ids are not anchors: (anchor, 2*c*c) resolved by the query compiler
                const uint32_t li = c.locf[f].list_begin;
                if (hit.present(li)) {
                    const float bv = as_global(reinterpret_cast<const float*>(c.lists[li].scores))[hit.index(li)];
                    *c.gb += 4u;
                    if (!have || bv < best) best = bv;
                    have = true;
                }
                continue;
            }
            uint32_t cnt = 0;
            for (uint32_t j = 0; j < c.locf[f].list_count; ++j) cnt += hit.present((uint32_t)c.loc_idx[c.locf[f].list_begin + j]) ? 1u : 0u;
            if (cnt > 1u) {
                float bv = 2.0f * (float)cnt * (float)cnt;
                if (!have || bv < best) best = bv;
                have = true;
            }
        }
        if (have) score *= best;
    }
    for (uint32_t f = 0; f < c.n_facets; ++f) {  // persistence.rs:164-175 count_values_for_ids
        const DFacet& fa = c.facets[f];
        if (doc >= fa.key_base && doc - fa.key_base < fa.num_keys) {
            const uint32_t row = doc - fa.key_base;
            if (fa.direct) {  // uniform: a scalar field
                const uint32_t v = as_global(fa.direct)[row];
                *c.gb += 4u;
                if (v < fa.num_values) facet_add(c.fc_keys, c.hist, fa.hist_off + v);
                continue;
            }
            const unsigned long long e0 = as_global(fa.offsets)[row], e1 = as_global(fa.offsets)[row + 1];
            *c.gb += 16u + 4u * (uint32_t)(e1 - e0);
            for (unsigned long long e = e0; e < e1; ++e) {
                const uint32_t v = as_global(fa.values)[e];
                if (v < fa.num_values) facet_add(c.fc_keys, c.hist, fa.hist_off + v);
            }
        }
    }
    return score;
}

// Score queue entries [0, count) (count <= 64: one per lane) and feed the top-k.
__device__ void tile_queue_flush(uint32_t count, const ScoreCtx& sc, const uint32_t* qdoc, const unsigned long long* qmask, const uint32_t* qidx,
                                 const CandState& cs, uint32_t top_k) {
    const uint32_t lane = threadIdx.x;
    const bool have = lane < count;
    unsigned long long key = 0ull;
    if (have) {
        const QueuedHit qh{qmask[lane], qidx, lane};
        const uint32_t doc = qdoc[lane];
        float score = tree_score_generic(sc, qh);
        score = sink_stages(sc, score, doc, qh);
        key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
    }
    bool pending = have && key > *cs.thr && key < cs.upper;
    while (true) {
        if (pending) {
            if (key > *cs.thr) {
                const uint32_t pos = atomicAdd(cs.n, 1u);
                if (pos < cs.cap) {
                    cs.cand[pos] = key;
                    pending = false;
                }
            } else pending = false;
        }
        const int need = __syncthreads_or(pending ? 1 : 0);
        if (!need) break;
        cand_prune(cs, top_k);
    }
}

// ------------------------------------------------------------------------------------ k_tile_scan
// LDS map (u32 units):
//   misc[8]: thr(2) cand_n hits_acc pad
//   cur[2][64] nxt[2][64]   per-list cursor / doc at the cursor, double buffered by tile parity
//   nxt_new[64] cnt_lo[64] cnt_hi[64]
//   desc[desc_cap/4]        the query descriptor (header, lists, programs, sink stages) staged from the blob
//   cand[2*cand_cap]        candidate keys (u64)
//   stack[stack_depth*256]  score stack, one column per thread
//   rootw[WW]  bm[(L+T)*WW] (list bitmaps, then the presence program's temporaries)  pre[L*WW] (u16)
constexpr uint32_t kLdsMisc = 0;
constexpr uint32_t kLdsCur = 8;
constexpr uint32_t kSurvCap = 256;
// the per-list arrays are sized to the launch's longest list table `ml` (even): cur[2][ml] nxt[2][ml] nxt_new[ml] cnt_lo[ml] cnt_hi[ml],
// then the compacted survivor codes u16[kSurvCap], then the descriptor
__host__ __device__ constexpr uint32_t lds_desc_off(uint32_t ml) { return kLdsCur + 7u * ml + kSurvCap / 2u; }

size_t tile_scan_lds_bytes(uint32_t n_bitmaps, uint32_t n_lists, uint32_t tile_words, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap, bool queue, uint32_t ml) {
    // (the facet counter cache, 2 * kFacetCache words, sits directly behind the descriptor: launches with facets pass desc_cap + its bytes)
    size_t u32s = lds_desc_off(ml) + desc_cap / 4 + 2 * (size_t)cand_cap + (size_t)stack_depth * kBlock + (size_t)tile_words + (size_t)n_bitmaps * tile_words +
                  ((size_t)n_lists * tile_words + 1) / 2;
    // survivor queue (queries of <= kQueueMaxLists lists): qmask u64[kQueueCap], qdoc u32[kQueueCap], qidx u32[n_lists][kQueueCap]
    if (queue && n_lists <= kQueueMaxLists) u32s = ((u32s + 1) & ~size_t(1)) + (size_t)kQueueCap * (3 + n_lists);
    return u32s * 4 + 16;
}

// one posting-list element of the tile scatter
#define VQ_SCATTER_ELEM(D, IDX)                              \
    if (!stop && (IDX) >= c0) {                              \
        if ((IDX) >= len || (D) >= tile_hi) {                \
            stop = true;                                     \
            if ((IDX) < len) first_ge = (D);                 \
        } else {                                             \
            ++nhi;                                           \
            if ((D) < tile_lo) ++nlo;                        \
            else {                                           \
                const uint32_t rel = (D)-tile_lo;            \
                atomicOr(&bmi[rel >> 5], 1u << (rel & 31u)); \
            }                                                \
        }                                                    \
    }

constexpr int kGroup = 4;  // lists whose first-round loads are issued together

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_tile_scan(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                      const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq,
                                                      uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap,
                                                      unsigned long long* __restrict__ span_keys,
                                                      unsigned long long* __restrict__ num_hits, uint32_t* __restrict__ hist, uint32_t queue_on, uint32_t ml) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // ---- which (query, span) is this workgroup?  (uniform)
    uint32_t q;
    {
        uint32_t lo = 0, hi = nq;
        const uint32_t wg = blockIdx.x;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (span_base[mid] <= wg) lo = mid;
            else hi = mid;
        }
        q = lo;
    }
    const uint32_t ql = q;
    q = qmap[ql];
    const uint32_t span = blockIdx.x - span_base[ql];

    // ---- stage the query descriptor into LDS: every later phase reads it from there, not from HBM
    const uint32_t kLdsDesc = lds_desc_off(ml);  // (ml: list-table size of the launch, even, >= every query's n_lists)
    uint32_t* desc = lds + kLdsDesc;
    const uint8_t* gblob8 = blobs + __builtin_amdgcn_readfirstlane(blob_off[q]);  // (in SGPRs: the interpreter's scalar reads go here)
    {
        const uint32_t* gblob = reinterpret_cast<const uint32_t*>(gblob8);
        const uint32_t n32 = reinterpret_cast<const QHeader*>(gblob)->desc_bytes >> 2;
        for (uint32_t x = tid; x < n32; x += kBlock) desc[x] = gblob[x];
    }
    __syncthreads();
    const uint8_t* blob = reinterpret_cast<const uint8_t*>(desc);
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const uint32_t L = H->n_lists;
    const uint32_t WW = H->tile_words;
    const uint32_t W = WW << 5;
    const uint32_t top_k = H->top_k;
    const DList* lists = reinterpret_cast<const DList*>(blob + H->off_lists);
    const DOp* ops = reinterpret_cast<const DOp*>(blob + H->off_ops);
    const DGroup* groups = reinterpret_cast<const DGroup*>(blob + H->off_groups);
    const DTermBoost* tboosts = reinterpret_cast<const DTermBoost*>(blob + H->off_tboost);
    const DColBoost* cols = reinterpret_cast<const DColBoost*>(blob + H->off_col);
    const DLocField* locf = reinterpret_cast<const DLocField*>(blob + H->off_locf);
    const uint16_t* loc_idx = reinterpret_cast<const uint16_t*>(blob + H->off_loc_idx);
    const DFacet* facets = reinterpret_cast<const DFacet*>(blob + H->off_facets);
    const DPresOp* pres = reinterpret_cast<const DPresOp*>(blob + H->off_pres);
    const uint16_t* pres_in = reinterpret_cast<const uint16_t*>(blob + H->off_pres_in);
    const VQ_CONST DOp* kops = as_const<DOp>(gblob8 + __builtin_amdgcn_readfirstlane(H->off_ops));
    const VQ_CONST DList* klists = as_const<DList>(gblob8 + __builtin_amdgcn_readfirstlane(H->off_lists));
    const uint32_t n_ops = H->n_ops, n_pres = H->n_pres;
    const uint32_t n_groups = H->n_groups, n_tboost = H->n_tboost, n_col = H->n_col, n_locf = H->n_locf, n_facets = H->n_facets;

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds + kLdsMisc);
    uint32_t* cand_n = lds + kLdsMisc + 2;
    uint32_t* hits_acc = lds + kLdsMisc + 3;
    uint32_t* cur2 = lds + kLdsCur;       // [2][ml]
    uint32_t* nxt2 = cur2 + 2u * ml;      // [2][ml]
    uint32_t* nxt_new = nxt2 + 2u * ml;
    uint32_t* cnt_lo = nxt_new + ml;
    uint32_t* cnt_hi = cnt_lo + ml;
    uint16_t* const surv_list = reinterpret_cast<uint16_t*>(cnt_hi + ml);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + kLdsDesc + desc_cap / 4);
    uint32_t* stack = lds + kLdsDesc + desc_cap / 4 + 2 * cand_cap + tid;  // this thread's column
    uint32_t* rootw = lds + kLdsDesc + desc_cap / 4 + 2 * cand_cap + stack_depth * kBlock;
    uint32_t* bm = rootw + WW;
    uint16_t* pre = reinterpret_cast<uint16_t*>(bm + (L + H->n_temps) * WW);
    CandState cs{cand, cand_n, thr, cand_cap,
                 reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blobs) + blob_off[q] + offsetof(QHeader, gthr))};
    cs.upper = H->key_upper;
    uint32_t tiles_done = 0;
    // survivor queue: the few docs per tile that the pruning leaves wait until 64 of them make a full scoring round (a round costs the
    // same for 1 and for 64 docs); their list memberships and posting indices are captured when they are queued
    const bool use_queue = (queue_on & 1u) && L <= kQueueMaxLists && !H->n_counts && !H->simple_n;  // uniform
    // facet counter cache: the last 2 * kFacetCache words of the descriptor area (the host sized desc_cap for it: queue_on bit 1)
    uint32_t* const fc_keys = ((queue_on & 2u) && n_facets && !H->n_counts) ? lds + kLdsDesc + desc_cap / 4 - 2u * kFacetCache : nullptr;
    if (fc_keys)
        for (uint32_t x = tid; x < kFacetCache; x += kBlock) {
            fc_keys[x] = 0xFFFFFFFFu;
            fc_keys[kFacetCache + x] = 0u;
        }
    uint32_t* const qbase = lds + ((((uint32_t)(reinterpret_cast<uint32_t*>(pre) - lds) + (L * WW + 1u) / 2u) + 1u) & ~1u);
    unsigned long long* const qmask = reinterpret_cast<unsigned long long*>(qbase);
    uint32_t* const qdoc = qbase + 2u * kQueueCap;
    uint32_t* const qidx = qdoc + kQueueCap;  // [L][kQueueCap]
    uint32_t qlen = 0;
    unsigned long long score_lists = 0ull;  // lists whose postings carry a value (wave-uniform: kept in SGPRs)
    if (use_queue) {
        for (uint32_t i = 0; i < L; ++i)
            if (lists[i].flags & LIST_HAS_SCORES) score_lists |= 1ull << i;
        score_lists = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(uint32_t)(score_lists >> 32)) << 32) |
                      (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)score_lists);
    }

    // ---- span of the doc-id space owned by this workgroup
    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    // span boundaries sit on tile boundaries (multiples of W), so a tile never straddles two workgroups
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & ~(W - 1u));
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & ~(W - 1u));
    const uint32_t bitmap_base = H->bitmap_base;
    const uint32_t keys_base = H->keys_base;

    VQ_STAMP_INIT
    VQ_STAMP_COUNT(8)
    // ---- initial cursors: first entry >= span_lo of every list, and the doc found there
    for (uint32_t i = wave; i < L; i += kBlock / 64) {
        const uint32_t len = lists[i].len;
        const uint32_t* docs = lists[i].docs;
        uint32_t c = wave_lower_bound(docs, len, span_lo);
        if (lane == 0) {
            cur2[i] = c;
            nxt2[i] = c < len ? as_global(docs)[c] : 0xFFFFFFFFu;
        }
    }
    if (tid == 0) {
        *thr = 0ull;
        *cand_n = 0;
        *hits_acc = 0;
    }
    const uint32_t n_counts = H->n_counts;
    for (uint32_t c = tid; c < n_counts; c += kBlock) reinterpret_cast<uint32_t*>(cand)[c] = 0u;  // the candidate area holds the counters
    uint32_t my_hits = 0;
    uint32_t my_gb = 0;  // bytes this lane read by per-hit gathers
    uint32_t par = 0;  // parity of the cursor buffers
    const bool seq_tiles = H->seq_tiles != 0u;
    uint32_t seq_pos = span_lo;
    __syncthreads();
    VQ_STAMP_AT(0)

    while (true) {
        uint32_t* cur = cur2 + par * ml;
        uint32_t* nxt = nxt2 + par * ml;
        // ---- P0: next tile = the tile holding the smallest pending doc of the cover lists (LDS only), or simply the next one
        uint32_t head = 0xFFFFFFFFu;
        if (seq_tiles) head = seq_pos;
        else
        for (uint32_t i = 0; i < L; ++i) {
            if (lists[i].flags & LIST_COVER) {
                const uint32_t d = nxt[i];
                head = d < head ? d : head;
            }
        }
        if (head >= span_hi) break;  // uniform: span exhausted
        const uint32_t tile_lo = head & ~(W - 1u);
        const uint32_t tile_end = tile_lo + W;  // may wrap to 0 at the top of the id space
        const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;
        const uint32_t lo_bound = tile_lo > span_lo ? tile_lo : span_lo;  // entries below it are not this tile's
        seq_pos = tile_end > tile_lo ? tile_end : 0xFFFFFFFFu;

        // ---- P0b: a list outside the cover that is more than a tile behind skips ahead with a wave-wide search
        for (uint32_t i = wave; i < L; i += kBlock / 64) {
            const uint32_t d = nxt[i];
            if (d < tile_lo && tile_lo - d >= W) {  // uniform per wave
                const uint32_t c = cur[i];
                uint32_t adv = wave_lower_bound(lists[i].docs + c, lists[i].len - c, tile_lo);
                if (lane == 0) cur[i] = c + adv;
            }
        }
        // ---- P1: clear the tile state (list bitmaps only: every temporary is fully written by its op)
        for (uint32_t x = tid * 4u; x < L * WW; x += kBlock * 4u) *reinterpret_cast<uint4*>(bm + x) = make_uint4(0u, 0u, 0u, 0u);
        if (tid < L) {
            cnt_lo[tid] = 0u;
            cnt_hi[tid] = 0u;
            nxt_new[tid] = 0xFFFFFFFFu;
        }
        if ((tiles_done++ & 3u) == 0u && tid == 0) {  // at the start and now and then: adopt the threshold other spans of the query have published
            const unsigned long long g = *reinterpret_cast<volatile unsigned long long*>(cs.gthr);
            if (g > *thr) *thr = g;
        }
        __syncthreads();
        VQ_STAMP_AT(1)
        VQ_STAMP_COUNT(7)

        // ---- P2: stream every list's doc ids of this tile into its bitmap.  One wave == one workgroup: each
        //      round loads 64 x 16 B (1 KiB, coalesced) of a list; how many of those 256 entries belong to the
        //      tile is counted with wave ballots (the list is sorted: the entries of the tile form a prefix), so
        //      the cursor bookkeeping needs no atomics.  First rounds of up to kGroup lists are issued together,
        //      later rounds are prefetched one ahead.
        const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (uint32_t g0 = 0; g0 < L; g0 += kGroup) {
            u32x4 first[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const uint32_t i = g0 + j;
                first[j] = kSent;
                if (i < L && (lists[i].flags & (LIST_BITMAP | LIST_COVER)) != LIST_BITMAP) {
                    const uint32_t v = (cur[i] >> 2) + lane;
                    if (v < ((lists[i].len + 3u) >> 2)) first[j] = as_global(reinterpret_cast<const u32x4*>(lists[i].docs))[v];
                }
            }
            VQ_STAMP_AT(9)
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
                const uint32_t i = g0 + j;
                if (i < L && (lists[i].flags & (LIST_BITMAP | LIST_COVER)) == LIST_BITMAP) {  // uniform
                    // dense list with a bitmap image in HBM: the tile is a straight 16 B/lane copy, ranks come
                    // from the rank directory (tiles are aligned to W >= 2048 docs)
                    const VQ_GLOBAL u32x4* gb = as_global(reinterpret_cast<const u32x4*>(lists[i].bitmap + ((tile_lo - bitmap_base) >> 5)));
                    for (uint32_t k = lane; k < (WW >> 2); k += 64u) reinterpret_cast<u32x4*>(bm + i * WW)[k] = gb[k];
                    if (lane == 0) {
                        cur[i] = as_global(lists[i].rank_dir)[(tile_lo - bitmap_base) >> kRankShift];
                        cnt_hi[i] = 0;
                        cnt_lo[i] = 0;
                        nxt_new[i] = 0xFFFFFFFFu;
                    }
                } else if (i < L) {  // uniform
                    // Entries before the cursor inside its 16-byte vector are < lo_bound (already consumed, or
                    // below this span), so counting from the vector-aligned cursor keeps every index consistent:
                    //   next cursor = c0v + total_in,  first in-tile entry = c0v + total_lo.
                    const uint32_t c0v = cur[i] & ~3u;
                    const uint32_t nvec = (lists[i].len + 3u) >> 2;
                    const VQ_GLOBAL u32x4* dptr = as_global(reinterpret_cast<const u32x4*>(lists[i].docs));
                    uint32_t* bmi = bm + i * WW;
                    uint32_t v = (c0v >> 2) + lane;
                    u32x4 d4 = first[j];
                    uint32_t total_in = 0, total_lo = 0, boundary = 0xFFFFFFFFu;
                    while (true) {  // uniform trip count
                        const uint32_t vn = v + 64u;
                        u32x4 nx = kSent;
                        if (vn < nvec) nx = dptr[vn];  // prefetch the next round
                        // padding and exhausted lanes hold the sentinel 0xFFFFFFFF (>= tile_hi): they end the list.
                        // The list is sorted: the in-tile entries of this round are a prefix in (lane, component) order.
                        const bool ix = d4.x < tile_hi, iy = d4.y < tile_hi, iz = d4.z < tile_hi, iw = d4.w < tile_hi;
                        const uint32_t mine = (uint32_t)ix + (uint32_t)iy + (uint32_t)iz + (uint32_t)iw;
                        const uint32_t full = (uint32_t)__popcll(__ballot(iw));  // lanes with all four entries inside
                        uint32_t n_in = full << 2;
                        if (full < 64u) n_in += (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)full);
                        total_in += n_in;
                        const unsigned long long lom = __ballot(d4.x < lo_bound);
                        if (lom) {  // rare: entries in front of the tile (a list that lags behind, or the span start)
                            const bool lx = d4.x < lo_bound, ly = d4.y < lo_bound, lz = d4.z < lo_bound, lw = d4.w < lo_bound;
                            total_lo += (uint32_t)(__popcll(lom) + __popcll(__ballot(ly)) + __popcll(__ballot(lz)) + __popcll(__ballot(lw)));
                            if (ix && !lx) atomicOr(&bmi[(d4.x - tile_lo) >> 5], 1u << ((d4.x - tile_lo) & 31u));
                            if (iy && !ly) atomicOr(&bmi[(d4.y - tile_lo) >> 5], 1u << ((d4.y - tile_lo) & 31u));
                            if (iz && !lz) atomicOr(&bmi[(d4.z - tile_lo) >> 5], 1u << ((d4.z - tile_lo) & 31u));
                            if (iw && !lw) atomicOr(&bmi[(d4.w - tile_lo) >> 5], 1u << ((d4.w - tile_lo) & 31u));
                        } else if (ix) {
                            // merge the bits of entries that share a bitmap word: one LDS atomic per distinct word
                            uint32_t wi = (d4.x - tile_lo) >> 5;
                            uint32_t m = 1u << ((d4.x - tile_lo) & 31u);
                            if (iy) {
                                const uint32_t w2 = (d4.y - tile_lo) >> 5, b2 = 1u << ((d4.y - tile_lo) & 31u);
                                if (w2 == wi) m |= b2;
                                else {
                                    atomicOr(&bmi[wi], m);
                                    wi = w2;
                                    m = b2;
                                }
                            }
                            if (iz) {
                                const uint32_t w2 = (d4.z - tile_lo) >> 5, b2 = 1u << ((d4.z - tile_lo) & 31u);
                                if (w2 == wi) m |= b2;
                                else {
                                    atomicOr(&bmi[wi], m);
                                    wi = w2;
                                    m = b2;
                                }
                            }
                            if (iw) {
                                const uint32_t w2 = (d4.w - tile_lo) >> 5, b2 = 1u << ((d4.w - tile_lo) & 31u);
                                if (w2 == wi) m |= b2;
                                else {
                                    atomicOr(&bmi[wi], m);
                                    wi = w2;
                                    m = b2;
                                }
                            }
                            atomicOr(&bmi[wi], m);
                        }
                        if (full < 64u) {
                            // lane `full` holds the first entry >= tile_hi: component index == its in-tile count
                            const uint32_t c = mine == 0 ? d4.x : mine == 1 ? d4.y : mine == 2 ? d4.z : d4.w;
                            boundary = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)full);
                            break;
                        }
                        d4 = nx;
                        v = vn;
                    }
                    if (lane == 0) {
                        cnt_hi[i] = total_in - (cur[i] & 3u);  // cursor advance
                        cnt_lo[i] = total_lo - (cur[i] & 3u);  // entries between the cursor and the first in-tile entry
                        nxt_new[i] = boundary;
                    }
                }
            }
            VQ_STAMP_AT(10)
        }
        __syncthreads();
        VQ_STAMP_AT(2)
        // cursors of the NEXT tile go to the other parity: nobody reads them before the next barrier,
        // and the score gathers below still see this tile's cursors
        if (tid < L) {
            cur2[(par ^ 1u) * ml + tid] = cur[tid] + cnt_hi[tid];
            nxt2[(par ^ 1u) * ml + tid] = nxt_new[tid];
        }

        // ---- P3: presence program (three-address code over bitmaps; the last op writes the root words).
        //      A lane owns WPL consecutive words in every op: no synchronisation between ops.
        const uint32_t WPL = WW >> 6;
        const uint32_t w0 = lane * WPL;
        uint32_t surv = 0;
        for (uint32_t o = 0; o < n_pres; ++o) {
            const uint32_t kind = pres[o].kind;
            const uint32_t n_in = pres[o].n_in;
            const uint16_t* in = pres_in + pres[o].in_begin;
            const uint32_t out = pres[o].out;
            if (kind == PRES_COUNT) {  // count pre-pass: hits of one node of the tree inside this tile
                const uint32_t ref = in[0];
                const uint32_t* src = bm + ((ref & kSlotTemp) ? L + (ref & 0x7FFFu) : ref) * WW + w0;
                uint32_t pc = 0;
                for (uint32_t k = 0; k < WPL; ++k) pc += (uint32_t)__popc(src[k]);
                uint32_t tot;
                (void)wave_excl_scan_u32(pc, &tot);
                if (lane == 0) reinterpret_cast<uint32_t*>(cand)[out] += tot;
                continue;
            }
            uint32_t* dst = (out == kSlotRoot ? rootw : bm + (L + (out & 0x7FFFu)) * WW) + w0;
            if ((WPL & 3u) == 0) {
                for (uint32_t k = 0; k < WPL; k += 4) {
                    uint4 v = kind == PRES_AND ? make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu) : make_uint4(0u, 0u, 0u, 0u);
                    for (uint32_t c = 0; c < n_in; ++c) {
                        const uint32_t ref = in[c];
                        const uint4 x = *reinterpret_cast<const uint4*>(bm + ((ref & kSlotTemp) ? L + (ref & 0x7FFFu) : ref) * WW + w0 + k);
                        if (kind == PRES_AND) {
                            v.x &= x.x; v.y &= x.y; v.z &= x.z; v.w &= x.w;
                        } else {
                            v.x |= x.x; v.y |= x.y; v.z |= x.z; v.w |= x.w;
                        }
                    }
                    if (kind == PRES_ZERO) v = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(dst + k) = v;
                    if (out == kSlotRoot) surv += (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w));
                }
            } else {
                for (uint32_t k = 0; k < WPL; ++k) {
                    uint32_t v = kind == PRES_AND ? 0xFFFFFFFFu : 0u;
                    for (uint32_t c = 0; c < n_in; ++c) {
                        const uint32_t ref = in[c];
                        const uint32_t x = bm[((ref & kSlotTemp) ? L + (ref & 0x7FFFu) : ref) * WW + w0 + k];
                        v = kind == PRES_AND ? (v & x) : (v | x);
                    }
                    if (kind == PRES_ZERO) v = 0u;
                    dst[k] = v;
                    if (out == kSlotRoot) surv += (uint32_t)__popc(v);
                }
            }
        }
        uint32_t S = 0;
        const uint32_t my_excl = wave_excl_scan_u32(surv, &S);
        VQ_STAMP_AT(3)

        uint32_t my_excl2 = my_excl;
#ifdef VQ_STAMP
        _acc[11] += S;  // hits of the tile
#endif
        if (S && !n_counts) {
            my_hits += surv;  // every doc of the root words is a hit, scored or not
            // ---- top-k pruning (exact): count, bit-sliced, how many of the score tree's leaf lists hold each doc; a doc in fewer than
            //      k_min of them cannot reach the threshold (QHeader::prune_gbits) and leaves the root words unscored
            const uint32_t prune_n = H->prune_n;
            const uint32_t thr_hi = (uint32_t)(*thr >> 32);
            if (prune_n && thr_hi) {  // uniform
                uint32_t k_min = 1;
                while (k_min <= prune_n && H->prune_gbits[k_min] < thr_hi) ++k_min;
                if (k_min > 1) {
                    const unsigned long long pmask = H->prune_mask;
                    surv = 0;
                    for (uint32_t k = 0; k < WPL; ++k) {
                        uint32_t rw = rootw[w0 + k];
                        if (rw) {
                            uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
                            for (uint32_t i = 0; i < L; ++i) {
                                if ((pmask >> i) & 1ull) {  // uniform
                                    const uint32_t x = bm[i * WW + w0 + k];
                                    const uint32_t c0 = p0 & x;
                                    p0 ^= x;
                                    const uint32_t c1 = p1 & c0;
                                    p1 ^= c0;
                                    const uint32_t c2 = p2 & c1;
                                    p2 ^= c1;
                                    p3 ^= c2;
                                }
                            }
                            uint32_t gt = 0u, eq = ~0u;  // count >= k_min, one bit per doc
                            {
                                const uint32_t pl[4] = {p0, p1, p2, p3};
#pragma unroll
                                for (int bit = 3; bit >= 0; --bit) {
                                    if ((k_min >> bit) & 1u) eq &= pl[bit];
                                    else {
                                        gt |= eq & pl[bit];
                                        eq &= ~pl[bit];
                                    }
                                }
                            }
                            rw = k_min > prune_n ? 0u : (rw & (gt | eq));  // (k_min == prune_n + 1: no doc can reach the threshold)
                            rootw[w0 + k] = rw;
                        }
                        surv += (uint32_t)__popc(rw);
                    }
                    my_excl2 = wave_excl_scan_u32(surv, &S);
                }
            }
        }
#ifdef VQ_STAMP
        _acc[12] += S;  // docs that survive the pruning (scored)
#endif
        if (S && !n_counts) {  // uniform
            // ---- P4: rank support for the score gathers: exclusive prefix popcount per posting list
            for (uint32_t i = 0; i < L; ++i) {
                if (lists[i].flags & LIST_HAS_SCORES) {
                    const uint32_t* words = bm + i * WW + w0;
                    uint32_t local = 0;
                    for (uint32_t k = 0; k < WPL; ++k) local += (uint32_t)__popc(words[k]);
                    uint32_t tot;
                    uint32_t run = wave_excl_scan_u32(local, &tot);
                    uint16_t* pw = pre + i * WW + w0;
                    for (uint32_t k = 0; k < WPL; ++k) {
                        pw[k] = (uint16_t)run;
                        run += (uint32_t)__popc(words[k]);
                    }
                }
            }
            // ---- P5: score the surviving docs, run the sink stages, feed the top-k.  Few survivors (AND):
            //      compact them so that every lane scores at most ceil(S/64); many (OR): each lane walks its own words.
            const bool compact = S <= kSurvCap;
            if (compact) {
                uint32_t pos = my_excl2;
                for (uint32_t k = 0; k < WPL; ++k) {
                    uint32_t r = rootw[w0 + k];
                    while (r) {
                        const uint32_t b = (uint32_t)__ffs((int)r) - 1u;
                        r &= r - 1u;
                        surv_list[pos++] = (uint16_t)(((w0 + k) << 5) | b);
                    }
                }
            }
            __syncthreads();  // one wave: orders the LDS writes above before the reads below
            VQ_STAMP_AT(4)

            ScoreCtx sc{lists, ops, n_ops, H->simple_n, groups, n_groups, tboosts, n_tboost, cols, n_col, locf, loc_idx, n_locf, facets, n_facets,
                        bm, pre, cur, cnt_lo, WW, reinterpret_cast<float*>(stack), hist, kops, klists, &my_gb, fc_keys};
            if (compact && use_queue) {
                for (uint32_t base = 0; base < S; base += 32u) {  // uniform
                    const uint32_t nb = S - base < 32u ? S - base : 32u;
                    if (lane < nb) {
                        const uint32_t code = surv_list[base + lane];
                        const uint32_t w = code >> 5, b = code & 31u, p = qlen + lane;
                        unsigned long long m = 0ull;
                        for (uint32_t li = 0; li < L; ++li) {  // uniform
                            const uint32_t word = bm[li * WW + w];
                            if ((word >> b) & 1u) {
                                m |= 1ull << li;
                                if ((score_lists >> li) & 1ull)
                                    qidx[li * kQueueCap + p] = cur[li] + cnt_lo[li] + (uint32_t)pre[li * WW + w] + (uint32_t)__popc(word & ((1u << b) - 1u));
                            }
                        }
                        qdoc[p] = tile_lo + (w << 5) + b;
                        qmask[p] = m;
                    }
                    qlen += nb;
                    __syncthreads();
                    if (qlen >= 64u) {  // uniform
                        tile_queue_flush(64u, sc, qdoc, qmask, qidx, cs, top_k);
                        const uint32_t rem = qlen - 64u;  // < 64: move it to the front (disjoint source and destination)
                        if (lane < rem) {
                            qdoc[lane] = qdoc[64u + lane];
                            qmask[lane] = qmask[64u + lane];
                            for (uint32_t li = 0; li < L; ++li)
                                if ((score_lists >> li) & 1ull) qidx[li * kQueueCap + lane] = qidx[li * kQueueCap + 64u + lane];
                        }
                        __syncthreads();
                        qlen = rem;
                    }
                }
            } else {
            uint32_t it_a = compact ? lane : 0u;
            uint32_t it_r = compact ? 0u : rootw[w0];
            bool pending = false;
            unsigned long long pend_key = 0ull;
            while (true) {
                while (true) {
                    if (pending) {
                        if (pend_key > *thr) {  // (pend_key < key_upper was checked when it was set aside)
                            uint32_t pos = atomicAdd(cand_n, 1u);
                            if (pos < cand_cap) {
                                cand[pos] = pend_key;
                                pending = false;
                            } else break;
                        } else pending = false;
                    }
                    uint32_t w, b;
                    if (compact) {
                        if (it_a >= S) break;
                        const uint32_t code = surv_list[it_a];
                        it_a += 64u;
                        w = code >> 5;
                        b = code & 31u;
                    } else {
                        while (it_r == 0u && ++it_a < WPL) it_r = rootw[w0 + it_a];
                        if (it_r == 0u) break;
                        b = (uint32_t)__ffs((int)it_r) - 1u;
                        it_r &= it_r - 1u;
                        w = w0 + it_a;
                    }
                    const uint32_t doc = tile_lo + (w << 5) + b;
#ifdef VQ_STAMP
                    const unsigned long long _ts0 = __builtin_amdgcn_s_memtime();
#endif
                    const TileHit th{sc, w, b};
                    float score = sc.simple_n ? tree_score_simple(sc, w, b) : tree_score_generic(sc, th);
#ifdef VQ_STAMP
                    _acc[13] += 1ull;                                      // scoring rounds of lane 0
                    _acc[14] += __builtin_amdgcn_s_memtime() - _ts0;       // ticks inside the score tree
#endif
                    score = sink_stages(sc, score, doc, th);
                    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
                    if (key > *thr && key < cs.upper) {
                        uint32_t pos = atomicAdd(cand_n, 1u);
                        if (pos < cand_cap) cand[pos] = key;
                        else {
                            pending = true;
                            pend_key = key;
                            break;
                        }
                    }
                }
                const int need = __syncthreads_or(pending ? 1 : 0);
                if (!need) break;
#ifdef VQ_STAMP
                _acc[15] += 1ull;  // candidate prunes inside P5
#endif
                cand_prune(cs, top_k);
            }
            }
            VQ_STAMP_AT(5)
        }
        par ^= 1u;
    }

    if (n_counts) {  // count pre-pass: publish this span's counters, nothing else
        __syncthreads();
        for (uint32_t c = tid; c < n_counts; c += kBlock) {
            const uint32_t v = reinterpret_cast<uint32_t*>(cand)[c];
            if (v) atomicAdd(&num_hits[H->part_keys_off + c], (unsigned long long)v);
        }
        return;
    }
    // ---- span done: score what is still queued, publish the local top-k (as a set) and the hit count
    if (qlen) {  // uniform
        __syncthreads();
        ScoreCtx sc{lists, ops, n_ops, H->simple_n, groups, n_groups, tboosts, n_tboost, cols, n_col, locf, loc_idx, n_locf, facets, n_facets,
                    bm, pre, cur2, cnt_lo, WW, reinterpret_cast<float*>(stack), hist, kops, klists, &my_gb, fc_keys};
        tile_queue_flush(qlen, sc, qdoc, qmask, qidx, cs, top_k);
    }
    cand_prune(cs, top_k);
    {
        const uint32_t n = *cand_n;
        unsigned long long* out = span_keys + (size_t)keys_base + (size_t)span * top_k;
        for (uint32_t i = tid; i < top_k; i += kBlock) out[i] = i < n ? cand[i] : 0ull;
    }
    if (my_hits) atomicAdd(hits_acc, my_hits);
    __syncthreads();
    if (fc_keys) facet_cache_flush(fc_keys, hist);
    if (tid == 0 && *hits_acc) atomicAdd(&num_hits[q], (unsigned long long)*hits_acc);
    {
        uint32_t gb_total;
        (void)wave_excl_scan_u32(my_gb, &gb_total);
        if (tid == 0 && gb_total && H->stat_off) atomicAdd(&num_hits[H->stat_off], (unsigned long long)gb_total);
    }
    VQ_STAMP_AT(6)
    VQ_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------ merges
constexpr uint32_t kMergeThreads = 1024;
__global__ __launch_bounds__(kMergeThreads) void k_merge_spans(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                               const unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ part_keys) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ unsigned long long lane_max[kBlock];
    __shared__ uint32_t misc[4];
    __shared__ uint32_t fill[2];  // [0] keys kept so far, [1] overflow flag
    const QHeader* H = reinterpret_cast<const QHeader*>(blobs + blob_off[blockIdx.x]);
    const uint32_t top_k = H->top_k;
    const unsigned long long* src = span_keys + H->keys_base;
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc), (uint32_t)kCandCap};
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t total = H->n_spans * top_k;
    bool rounds = true;  // uniform
    if (top_k >= 1u && top_k <= (uint32_t)kBlock && total > 4u * (uint32_t)kBlock) {
        // few keys wanted out of many: the top_k-th largest of 64 maxima over disjoint subsets is a lower bound of the answer's smallest
        // key (at least top_k keys reach it), so a second pass that keeps only the keys at or above it leaves a handful to sort
        if (tid < (uint32_t)kBlock) lane_max[tid] = 0ull;
        if (tid == 0) fill[0] = fill[1] = 0u;
        __syncthreads();
        unsigned long long mx = 0ull;
#pragma unroll 4
        for (uint32_t i = tid; i < total; i += kMergeThreads) {
            const unsigned long long v = src[i];
            mx = v > mx ? v : mx;
        }
        atomicMax(&lane_max[lane], mx);
        __syncthreads();
        const unsigned long long mine = lane_max[lane];
        uint32_t rank = 0;
        for (uint32_t l = 0; l < (uint32_t)kBlock; ++l) {
            const unsigned long long v = lane_max[l];
            rank += (v > mine || (v == mine && l < lane)) ? 1u : 0u;
        }
        const unsigned long long pick = __ballot(rank == top_k - 1u);  // (every wave computes the same)
        unsigned long long bound = lane_max[(uint32_t)__ffsll((long long)pick) - 1u];
        if (bound == 0ull) bound = 1ull;  // empty slots never enter
        for (uint32_t i0 = 0; i0 < total; i0 += 4u * kMergeThreads) {  // uniform trip count; four loads in flight
            unsigned long long v4[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) v4[j] = i0 + j * kMergeThreads + tid < total ? src[i0 + j * kMergeThreads + tid] : 0ull;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const unsigned long long v = v4[j];
                const bool keep = v >= bound;
                const unsigned long long m = __ballot(keep);
                if (m) {
                    const uint32_t cnt = (uint32_t)__popcll(m);
                    uint32_t base = 0;
                    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&fill[0], cnt);
                    base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
                    if (base + cnt > (uint32_t)kCandCap) fill[1] = 1u;  // (only when nearly all keys tie: take the general path below)
                    else if (keep) cand[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
                }
            }
        }
        __syncthreads();
        rounds = fill[1] != 0u;
        if (!rounds) {
            if (tid >= (uint32_t)kBlock) return;  // one wave finishes (cand_prune is written for kBlock threads)
            if (tid == 0) {
                *cs.thr = 0ull;
                *cs.n = fill[0];
            }
            cand_prune(cs, top_k, true);
        }
    }
    if (rounds) {
        if (tid >= (uint32_t)kBlock) return;
        if (tid == 0) {
            *cs.thr = 0ull;
            *cs.n = 0;
        }
        __syncthreads();
        // fill the buffer with as many spans as fit next to the kept top_k, prune, continue
        const uint32_t per_round = ((uint32_t)kCandCap - top_k) / top_k;
        for (uint32_t s0 = 0; s0 < H->n_spans; s0 += per_round) {
            const uint32_t cnt = (H->n_spans - s0 < per_round ? H->n_spans - s0 : per_round) * top_k;
            const uint32_t base = *cs.n;
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += kBlock) cand[base + i] = src[(size_t)s0 * top_k + i];
            __syncthreads();
            if (tid == 0) *cs.n = base + cnt;
            cand_prune(cs, top_k, true);  // sorted descending: empty slots (0) sink to the end
        }
    }
    unsigned long long* out = part_keys + H->part_keys_off;
    const uint32_t n = *cs.n;
    for (uint32_t i = tid; i < top_k; i += kBlock) out[i] = i < n ? cand[i] : 0ull;
}

// gathered: the all-gathered parts (hit counts, counters, keys: PartialLayout::off_hist bytes each) of num_shards packed partial buffers, shard-major.
__global__ __launch_bounds__(kBlock) void k_finalize(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                     const uint8_t* __restrict__ gathered, uint32_t num_shards, size_t shard_stride, PartialLayout lay,
                                                     uint32_t* __restrict__ res_ids, float* __restrict__ res_scores, uint32_t* __restrict__ res_n,
                                                     unsigned long long* __restrict__ res_hits) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ uint32_t misc[4];
    const uint32_t q = blockIdx.x;
    const QHeader* H = reinterpret_cast<const QHeader*>(blobs + blob_off[q]);
    const uint32_t top_k = H->top_k;
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc), (uint32_t)kCandCap};
    if (threadIdx.x == 0) {
        *cs.thr = 0ull;
        *cs.n = 0;
    }
    __syncthreads();
    unsigned long long hits = 0;
    for (uint32_t s = 0; s < num_shards; ++s) {
        const uint8_t* pb = gathered + (size_t)s * shard_stride;
        const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(pb + lay.off_keys) + H->part_keys_off;
        hits += reinterpret_cast<const unsigned long long*>(pb + lay.off_hits)[q];
        if (*cs.n + top_k > (uint32_t)kCandCap) cand_prune(cs, top_k);  // uniform: *cs.n is stable here
        const uint32_t base = *cs.n;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) cand[base + i] = keys[i];
        __syncthreads();
        if (threadIdx.x == 0) *cs.n = base + top_k;
        __syncthreads();
    }
    cand_prune(cs, top_k, true);  // final ranking: (score desc, id desc); empty slots (0) last
    const uint32_t n = *cs.n;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) {
        const unsigned long long k = i < n ? cand[i] : 0ull;
        res_ids[H->part_keys_off + i] = (uint32_t)(k & 0xFFFFFFFFull);
        res_scores[H->part_keys_off + i] = __uint_as_float(unorder_f32((uint32_t)(k >> 32)));
        mine += k != 0ull ? 1u : 0u;
    }
    // number of real hits in the top-k window: the non-empty keys (unique, non-zero, sorted first).  min(total hits, top_k) without a
    // key bound; fewer when only keys below QHeader::key_upper were ranked (a later page of a deep request)
    uint32_t ranked;
    (void)wave_excl_scan_u32(mine, &ranked);
    if (threadIdx.x == 0) {
        res_hits[q] = hits;
        res_n[q] = ranked;
    }
}

// A few entries out of a long histogram (top 10 of 65 536 tag values) is a streaming job: k_facet_select_wide takes those with four
// waves and 16-byte loads, k_facet_select the rest.
__device__ __forceinline__ bool facet_job_is_wide(const FacetJob& job) { return job.top >= 1u && job.top <= 64u && job.num_values >= 4096u; }

constexpr uint32_t kFacetWideThreads = 256;
constexpr uint32_t kFacetWideCap = 1024;  // keys at or above the bound that fit in LDS
// One workgroup of four waves per job.  Pass 1 streams the counters (16 B per lane and load, eight loads in flight) and keeps every thread's
// largest key; the k-th largest of the 256 thread maxima is a lower bound of the answer's smallest key (at least k keys reach it).  Pass 2 reads the counters again (from L2 / Infinity Cache) and collects the keys at or above the bound — a
// handful; should more than kFacetWideCap reach it (one thread held all the large counters), the bound is raised to the k-th largest of those
// collected and the pass repeats.  The collected keys are ranked by counting (keys are unique: count and value id).
__global__ __launch_bounds__(kFacetWideThreads) void k_facet_select_wide(const FacetJob* __restrict__ jobs, const uint32_t* __restrict__ hist,
                                                                         uint32_t* __restrict__ out_vals, uint32_t* __restrict__ out_counts,
                                                                         uint32_t* __restrict__ out_n) {
    __shared__ unsigned long long cand[kFacetWideCap];
    __shared__ unsigned long long wbound[1];
    __shared__ uint32_t cn;
    const FacetJob job = jobs[blockIdx.x];
    if (!facet_job_is_wide(job)) return;  // uniform
    const uint32_t tid = threadIdx.x;
    const uint32_t k = job.top, nv = job.num_values;
    const uint32_t* h = hist + job.hist_off;
    const uint32_t lead = (4u - (job.hist_off & 3u)) & 3u;  // counters in front of the first 16-byte boundary (the batch's histograms lie back to back)
    const uint32_t nvec = (nv - lead) >> 2, tail0 = lead + (nvec << 2);
    const VQ_GLOBAL u32x4* h4 = as_global(reinterpret_cast<const u32x4*>(h + lead));
    auto key_of = [](uint32_t count, uint32_t v) { return ((unsigned long long)count << 32) | (unsigned long long)(0xFFFFFFFFu - v); };  // count desc, value id asc
    constexpr uint32_t kU = 8;
    unsigned long long mx = 0ull;
    for (uint32_t base = 0; base < nvec; base += kU * kFacetWideThreads) {  // uniform
        u32x4 c[kU];
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) {
            const uint32_t v = base + j * kFacetWideThreads + tid;
            c[j] = v < nvec ? h4[v] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (uint32_t j = 0; j < kU; ++j) {
            const uint32_t v = lead + (base + j * kFacetWideThreads + tid) * 4u;
            const uint32_t cc[4] = {c[j].x, c[j].y, c[j].z, c[j].w};
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e)
                if (cc[e]) {
                    const unsigned long long key = key_of(cc[e], v + e);
                    mx = key > mx ? key : mx;
                }
        }
    }
    for (uint32_t i = tid; i < lead + (nv - tail0); i += kFacetWideThreads) {  // (the counters in front of and behind the 16-byte vectors)
        const uint32_t v = i < lead ? i : tail0 + (i - lead);
        if (h[v]) {
            const unsigned long long key = key_of(h[v], v);
            mx = key > mx ? key : mx;
        }
    }
    // the k-th largest of the 256 thread maxima (ranked by counting; zero maxima — threads without a non-zero counter — tie and are told apart by thread)
    cand[tid] = mx;
    if (tid == 0) wbound[0] = 0ull;
    __syncthreads();
    {
        uint32_t rank = 0;
        for (uint32_t t = 0; t < kFacetWideThreads; ++t) {
            const unsigned long long o = cand[t];
            rank += (o > mx || (o == mx && t < tid)) ? 1u : 0u;
        }
        if (rank == k - 1u) wbound[0] = mx;
    }
    __syncthreads();
    unsigned long long bound = wbound[0] > 1ull ? wbound[0] : 1ull;  // (zero counts never enter)
    __syncthreads();
    while (true) {  // uniform
        if (tid == 0) cn = 0u;
        __syncthreads();
        auto push = [&](uint32_t count, uint32_t v) {
            if (count) {
                const unsigned long long key = key_of(count, v);
                if (key >= bound) {
                    const uint32_t pos = atomicAdd(&cn, 1u);
                    if (pos < kFacetWideCap) cand[pos] = key;
                }
            }
        };
        for (uint32_t base = 0; base < nvec; base += kU * kFacetWideThreads) {  // uniform
            u32x4 c[kU];
#pragma unroll
            for (uint32_t j = 0; j < kU; ++j) {
                const uint32_t v = base + j * kFacetWideThreads + tid;
                c[j] = v < nvec ? h4[v] : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (uint32_t j = 0; j < kU; ++j) {
                const uint32_t v = lead + (base + j * kFacetWideThreads + tid) * 4u;
                if ((unsigned long long)(c[j].x | c[j].y | c[j].z | c[j].w) << 32 >= (bound & 0xFFFFFFFF00000000ull)) {  // (some count of the four may reach the bound's)
                    push(c[j].x, v);
                    push(c[j].y, v + 1u);
                    push(c[j].z, v + 2u);
                    push(c[j].w, v + 3u);
                }
            }
        }
        for (uint32_t i = tid; i < lead + (nv - tail0); i += kFacetWideThreads) {
            const uint32_t v = i < lead ? i : tail0 + (i - lead);
            push(h[v], v);
        }
        __syncthreads();
        const uint32_t n = cn < kFacetWideCap ? cn : kFacetWideCap;
        // rank by counting: thread t takes keys t, t + 256, ...; keys are unique
        const bool fits = cn <= kFacetWideCap;  // uniform
        for (uint32_t i = tid; i < n; i += kFacetWideThreads) {
            const unsigned long long key = cand[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) rank += cand[j] > key ? 1u : 0u;
            if (fits) {
                if (rank < k) {
                    out_vals[job.out_off + rank] = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
                    out_counts[job.out_off + rank] = (uint32_t)(key >> 32);
                }
            } else if (rank == k - 1u) wbound[0] = key;  // the k-th largest of the keys collected (exactly one thread holds it): 1024 distinct keys reach the old bound, so it lies above it
        }
        if (fits) {
            for (uint32_t i = n + tid; i < k; i += kFacetWideThreads) {
                out_vals[job.out_off + i] = 0xFFFFFFFFu;
                out_counts[job.out_off + i] = 0u;
            }
            if (tid == 0) out_n[blockIdx.x] = n < k ? n : k;
            return;
        }
        __syncthreads();
        bound = wbound[0];  // strictly tighter every round: the loop ends
        __syncthreads();
    }
}

// one workgroup per facet entry of the batch
__global__ __launch_bounds__(kBlock) void k_facet_select(const FacetJob* __restrict__ jobs, const uint32_t* __restrict__ hist,
                                                         uint32_t* __restrict__ out_vals, uint32_t* __restrict__ out_counts,
                                                         uint32_t* __restrict__ out_n) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ uint32_t misc[4];
    const FacetJob job = jobs[blockIdx.x];
    if (job.top == 0u) {  // more entries wanted than the candidate buffer ranks: the host selects from the histogram itself (finish_batch)
        if (threadIdx.x == 0) out_n[blockIdx.x] = 0u;
        return;
    }
    if (facet_job_is_wide(job)) return;  // k_facet_select_wide's
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc), (uint32_t)kCandCap};
    if (threadIdx.x == 0) {
        *cs.thr = 0ull;
        *cs.n = 0;
    }
    __syncthreads();
    const uint32_t* h = hist + job.hist_off;
    const uint32_t k = job.top;
    // Few entries wanted out of many (top 10 of 65 536 tag values): pass 1 keeps each lane's largest key in a register; the k-th largest of the 64
    // lane maxima is a lower bound of the answer's smallest key (at least k keys reach it), so pass 2 pushes only the keys at or above it — a
    // handful — instead of feeding every non-zero count through the candidate buffer and its prunes.  kFacetPerRound loads are in flight together.
    constexpr uint32_t kFacetPerRound = 16;
    auto key_of = [](uint32_t count, uint32_t v) { return ((unsigned long long)count << 32) | (unsigned long long)(0xFFFFFFFFu - v); };  // count desc, value id asc
    unsigned long long bound = 1ull;  // (zero counts never enter)
    if (k >= 1u && k <= (uint32_t)kBlock && job.num_values > 8u * (uint32_t)kBlock) {
        unsigned long long mx = 0ull;
        for (uint32_t base = 0; base < job.num_values; base += kFacetPerRound * kBlock) {
            uint32_t c[kFacetPerRound];
#pragma unroll
            for (uint32_t j = 0; j < kFacetPerRound; ++j) {
                const uint32_t v = base + j * kBlock + threadIdx.x;
                c[j] = v < job.num_values ? h[v] : 0u;
            }
#pragma unroll
            for (uint32_t j = 0; j < kFacetPerRound; ++j)
                if (c[j]) {
                    const unsigned long long key = key_of(c[j], base + j * kBlock + threadIdx.x);
                    mx = key > mx ? key : mx;
                }
        }
        uint32_t rank = 0;  // position of this lane's maximum among the 64 (ties: lower lane first)
        for (uint32_t l = 0; l < (uint32_t)kBlock; ++l) {
            const unsigned long long o = shfl_u64(mx, l);
            rank += (o > mx || (o == mx && l < threadIdx.x)) ? 1u : 0u;
        }
        const unsigned long long pick = __ballot(rank == k - 1u);
        const unsigned long long kth = shfl_u64(mx, (uint32_t)__ffsll((long long)pick) - 1u);
        if (kth > bound) bound = kth;
    }
    for (uint32_t base = 0; base < job.num_values; base += kFacetPerRound * kBlock) {  // uniform trip count
        uint32_t c[kFacetPerRound];
        bool any = false;
#pragma unroll
        for (uint32_t j = 0; j < kFacetPerRound; ++j) {
            const uint32_t v = base + j * kBlock + threadIdx.x;
            c[j] = v < job.num_values ? h[v] : 0u;
            any = any || (c[j] && key_of(c[j], v) >= bound);
        }
        if (!__syncthreads_or(any)) continue;  // uniform: nothing of this round can enter
#pragma unroll
        for (uint32_t j = 0; j < kFacetPerRound; ++j) {
            const uint32_t v = base + j * kBlock + threadIdx.x;
            const unsigned long long key = c[j] ? key_of(c[j], v) : 0ull;
            bool pending = key >= bound && key > *cs.thr;
            if (!__syncthreads_or(pending)) continue;  // uniform
            while (true) {
                if (pending) {
                    uint32_t pos = atomicAdd(cs.n, 1u);
                    if (pos < (uint32_t)kCandCap) {
                        cand[pos] = key;
                        pending = false;
                    }
                }
                const int need = __syncthreads_or(pending ? 1 : 0);
                if (!need) break;
                cand_prune(cs, k);
                if (pending && !(key > *cs.thr)) pending = false;
            }
        }
    }
    cand_prune(cs, k, true);
    const uint32_t n = *cs.n;
    for (uint32_t i = threadIdx.x; i < k; i += kBlock) {
        const unsigned long long key = i < n ? cand[i] : 0ull;
        out_vals[job.out_off + i] = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
        out_counts[job.out_off + i] = (uint32_t)(key >> 32);
    }
    if (threadIdx.x == 0) out_n[blockIdx.x] = n;
}

// ------------------------------------------------------------------------------------ launchers
void launch_tile_scan(hipStream_t st, uint32_t total_spans, size_t lds_bytes, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                      const uint32_t* qmap, uint32_t nq, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap, unsigned long long* span_keys,
                      unsigned long long* num_hits, uint32_t* hist, bool queue, uint32_t ml, bool facet_cache) {
    if (!total_spans) return;
    hipLaunchKernelGGL(k_tile_scan, dim3(total_spans), dim3(kBlock), lds_bytes, st, blobs, blob_off, span_base, qmap, nq, stack_depth, cand_cap,
                       desc_cap, span_keys, num_hits, hist, (queue ? 1u : 0u) | (facet_cache ? 2u : 0u), ml);
}
void launch_merge_spans(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const unsigned long long* span_keys,
                        unsigned long long* part_keys) {
    if (!nq) return;
    hipLaunchKernelGGL(k_merge_spans, dim3(nq), dim3(kMergeThreads), 0, st, blobs, blob_off, span_keys, part_keys);
}
void launch_finalize(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const uint8_t* gathered, uint32_t num_shards,
                     size_t shard_stride, const PartialLayout& lay, uint32_t* res_ids, float* res_scores, uint32_t* res_n, unsigned long long* res_hits) {
    if (!nq) return;
    hipLaunchKernelGGL(k_finalize, dim3(nq), dim3(kBlock), 0, st, blobs, blob_off, gathered, num_shards, shard_stride, lay, res_ids, res_scores, res_n, res_hits);
}
// self-check (tests): the top `top` entries of one histogram through the select kernels -> number of entries written
int debug_facet_select(const uint32_t* hist_host, uint32_t num_values, uint32_t top, uint32_t misalign, uint32_t* out_vals_host, uint32_t* out_counts_host) {
    if (top == 0 || top > (uint32_t)kMaxTopK || misalign > 3u) return -1;
    uint32_t *d_hist = nullptr, *d_vals = nullptr, *d_counts = nullptr, *d_n = nullptr;
    FacetJob* d_job = nullptr;
    const FacetJob job{misalign, num_values, top, 0u};  // (the histogram starts `misalign` counters behind a 16-byte boundary)
    uint32_t n = 0;
    bool ok = hipMalloc(&d_hist, (size_t)num_values * 4 + 32) == hipSuccess && hipMalloc(&d_vals, (size_t)top * 4) == hipSuccess && hipMalloc(&d_counts, (size_t)top * 4) == hipSuccess &&
              hipMalloc(&d_n, 4) == hipSuccess && hipMalloc(&d_job, sizeof job) == hipSuccess;
    ok = ok && hipMemcpy(d_hist + misalign, hist_host, (size_t)num_values * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_job, &job, sizeof job, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        launch_facet_select(nullptr, 1, d_job, d_hist, d_vals, d_counts, d_n);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(&n, d_n, 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(out_vals_host, d_vals, (size_t)top * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(out_counts_host, d_counts, (size_t)top * 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_hist);
    (void)hipFree(d_vals);
    (void)hipFree(d_counts);
    (void)hipFree(d_n);
    (void)hipFree(d_job);
    return ok ? (int)n : -1;
}
void launch_facet_select(hipStream_t st, uint32_t n_jobs, const FacetJob* jobs, const uint32_t* hist, uint32_t* out_vals, uint32_t* out_counts,
                         uint32_t* out_n) {
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_facet_select, dim3(n_jobs), dim3(kBlock), 0, st, jobs, hist, out_vals, out_counts, out_n);
    hipLaunchKernelGGL(k_facet_select_wide, dim3(n_jobs), dim3(kFacetWideThreads), 0, st, jobs, hist, out_vals, out_counts, out_n);
}

}  // namespace vq

// ====================================================================================================
// k_scan_simple — the scan for "pure simple" queries: 1..4 single-list posting leaves under one AND / OR
// (or a single leaf), no filter and no sink stages.  This is the shape of the headline workloads
// (single-term scan, 3-term AND, 3-term OR), so it gets its own kernel with everything that does not
// change from tile to tile in registers:
//   * tile of NV * 8192 docs (NV = 2): every lane owns NV * 4 consecutive bitmap words (NV 16-byte vectors)
//   * dense lists are read as bitmap words straight from HBM into registers (no LDS round trip);
//     sparse lists are scattered into an LDS bitmap with the ballot-counted cursor logic
//   * presence = register AND/OR of the word vectors; ranks = lane-local popcounts + one DPP scan per list
//   * survivors are appended to an LDS queue and scored 64 at a time: all score gathers of a flush are
//     in flight together (an AND tile usually has far fewer than 64 survivors)
// Same results as k_tile_scan bit for bit (tests run every simple query through both kernels).
// ====================================================================================================
namespace vq {

constexpr uint32_t kSW = 8192;   // docs per tile
constexpr uint32_t kSWW = 256;   // bitmap words per tile
constexpr uint32_t kQCap = 128;  // survivor queue entries
// LDS map (u32): misc[8] | ub[16] (rich queries: DSimple2::ub) | qdoc[kQCap] | qidx[4][kQCap] | qmask[kQCap] | cand[2*cand_cap] | bm[scattered lists][SWW]
constexpr uint32_t kSLdsUb = 8;
constexpr uint32_t kSLdsQDoc = kSLdsUb + 16;
constexpr uint32_t kSLdsQIdx = kSLdsQDoc + kQCap;
constexpr uint32_t kSLdsQMask = kSLdsQIdx + 4 * kQCap;  // rich queries: side-list membership bits of the queued doc
constexpr uint32_t kSLdsCand = kSLdsQMask + kQCap;

size_t scan_simple_lds_bytes(uint32_t cand_cap, uint32_t nv, uint32_t n_scatter, bool facet_cache) {
    return (size_t)(kSLdsCand + 2 * cand_cap + n_scatter * kSWW * nv + (facet_cache ? 2 * kFacetCache : 0)) * 4 + 16;
}

struct SimpleLeaf {
    const uint32_t* docs;
    const uint16_t* scores;
    const uint32_t* bitmap;
    const uint32_t* rank_dir;
    uint32_t len;
    float ts;
};

__device__ __forceinline__ uint32_t popc4(const u32x4& v) { return (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w)); }
__device__ __forceinline__ uint32_t comp4(const u32x4& v, uint32_t j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// Score queue entries [0, count) (count <= 64), push the keys into the candidate buffer.
// `stat` != null: the flush's gathered bytes are added to that LDS word (profiling, QHeader::stat_off)
__device__ void simple_flush(uint32_t count, uint32_t n, uint32_t kind, const SimpleLeaf (&lf)[4], const uint8_t (&order)[4], const uint8_t (&slot)[4],
                             uint32_t nslots, const uint32_t* qdoc, const uint32_t* qidx, const CandState& cs, uint32_t top_k, uint32_t* stat) {
    const uint32_t lane = threadIdx.x;
    const bool have = lane < count;
    uint32_t doc = 0;
    uint32_t idx[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (have) {
        doc = qdoc[lane];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n) idx[k] = qidx[k * kQCap + lane];
    }
    uint16_t raw[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n && idx[k] != 0xFFFFFFFFu) raw[k] = as_global(lf[k].scores)[idx[k]];
    if (stat) {  // uniform
        uint32_t g = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n) g += 2u * (uint32_t)__popcll(__ballot(idx[k] != 0xFFFFFFFFu));
        if (lane == 0) *stat += g;
    }
    float val[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) val[k] = posting_value(lf[k].ts, raw[k]);
    float score;
    if (n == 1) score = val[0];
    else if (kind == OP_AND) {  // set_op.rs:415-416: others summed first, the shortest list's score last
        score = 0.0f;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n) score += pick4(val[0], val[1], val[2], val[3], order[k]);
    } else {  // set_op.rs:169-186
        float sum = 0.0f, nd = 0.0f;
        for (uint32_t s = 0; s < nslots; ++s) {
            float m = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < n && slot[k] == s && idx[k] != 0xFFFFFFFFu) m = fmaxf(m, val[k]);
            if (m >= 0.00001f) nd += 1.0f;
            sum += m;
        }
        score = sum * nd * nd;
    }
    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
    bool pending = have && key > *cs.thr && key < cs.upper;
    while (true) {
        if (pending) {
            if (key > *cs.thr) {
                uint32_t pos = atomicAdd(cs.n, 1u);
                if (pos < cs.cap) {
                    cs.cand[pos] = key;
                    pending = false;
                }
            } else pending = false;
        }
        const int need = __syncthreads_or(pending ? 1 : 0);
        if (!need) break;
        cand_prune(cs, top_k);
    }
}

// ---- rich simple queries: shape of the tree and of the sink stages, all wave-uniform
struct RichShape {
    uint32_t ngroups, root_kind, root_nslots, n_side;
    uint32_t g_kind4, g_mask4, g_nslots4, r_order4, r_slot4;
    uint32_t g_order4[4], g_slot4[4];
    uint32_t n_grp, n_tb, n_loc, grp_mask4, tb_side4, loc_leaf2, loc_side2, has_filter, filter_mask, f32_mask;
    const DFacet* facets;
    uint32_t n_facets;
    uint32_t* hist;
    uint32_t* fc_keys;  // facet counter cache in LDS (null: none)
    float grp_mult[4], tb_mult[4];
    const DColBoost* cols;
    uint32_t n_col;
    uint32_t prune;  // DSimple2::prune: docs whose bound (DSimple2::ub, in LDS) lies below the threshold are counted, not queued
};
__device__ __forceinline__ uint32_t b8(uint32_t x, uint32_t i) { return (x >> (8u * i)) & 0xFFu; }

__device__ __forceinline__ RichShape load_rich_shape(const DSimple2* S2, const DColBoost* cols, uint32_t n_col) {
    RichShape R;
    R.ngroups = S2->ngroups; R.root_kind = S2->root_kind; R.root_nslots = S2->root_nslots; R.n_side = S2->n_side;
    R.g_kind4 = *reinterpret_cast<const uint32_t*>(S2->g_kind);
    R.g_mask4 = *reinterpret_cast<const uint32_t*>(S2->g_mask);
    R.g_nslots4 = *reinterpret_cast<const uint32_t*>(S2->g_nslots);
    R.r_order4 = *reinterpret_cast<const uint32_t*>(S2->r_order);
    R.r_slot4 = *reinterpret_cast<const uint32_t*>(S2->r_slot);
#pragma unroll
    for (uint32_t g = 0; g < 4; ++g) {
        R.g_order4[g] = *reinterpret_cast<const uint32_t*>(S2->g_order[g]);
        R.g_slot4[g] = *reinterpret_cast<const uint32_t*>(S2->g_slot[g]);
        R.grp_mult[g] = S2->grp_mult[g];
        R.tb_mult[g] = S2->tb_mult[g];
    }
    R.n_grp = S2->n_grp; R.n_tb = S2->n_tb; R.n_loc = S2->n_loc; R.has_filter = S2->has_filter; R.filter_mask = S2->filter_mask; R.f32_mask = S2->f32_mask;
    R.facets = nullptr; R.n_facets = 0; R.hist = nullptr; R.fc_keys = nullptr;
    R.grp_mask4 = *reinterpret_cast<const uint32_t*>(S2->grp_mask);
    R.tb_side4 = *reinterpret_cast<const uint32_t*>(S2->tb_side);
    R.loc_leaf2 = (uint32_t)S2->loc_leaf[0] | ((uint32_t)S2->loc_leaf[1] << 8);
    R.loc_side2 = (uint32_t)S2->loc_side[0] | ((uint32_t)S2->loc_side[1] << 8);
    R.cols = cols;
    R.n_col = n_col;
    R.prune = S2->prune;
    return R;
}

// Score queue entries [0, count) of a rich simple query: leaves -> groups -> root (same arithmetic and order as tree_score_generic),
// then the sink stages in the reference's order (column boosts, phrase groups, term boosts, text locality).
__device__ void rich_flush(uint32_t count, uint32_t n, const SimpleLeaf (&lf)[4], const RichShape& R, const uint32_t* qdoc, const uint32_t* qidx,
                           const uint32_t* qmask, const CandState& cs, uint32_t top_k, uint32_t* stat) {
    const uint32_t lane = threadIdx.x;
    const bool have = lane < count;
    uint32_t doc = 0, qm = 0, pm = 0;
    uint32_t idx[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (have) {
        doc = qdoc[lane];
        qm = qmask[lane];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n) idx[k] = qidx[k * kQCap + lane];
    }
    uint32_t raw[4] = {0, 0, 0, 0};
    uint32_t gb = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n && idx[k] != 0xFFFFFFFFu) {
            if ((R.f32_mask >> k) & 1u) raw[k] = as_global(reinterpret_cast<const uint32_t*>(lf[k].scores))[idx[k]];  // uniform: materialised leaf
            else raw[k] = as_global(lf[k].scores)[idx[k]];
            gb += ((R.f32_mask >> k) & 1u) ? 4u : 2u;
            pm |= 1u << k;
        }
    float val[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) val[k] = ((R.f32_mask >> k) & 1u) ? __uint_as_float(raw[k]) : posting_value(lf[k].ts, (uint16_t)raw[k]);
    // groups
    float gv[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t gp = 0;
#pragma unroll
    for (uint32_t g = 0; g < 4; ++g) {
        if (g < R.ngroups) {  // uniform
            const uint32_t gk = b8(R.g_kind4, g), gm = b8(R.g_mask4, g);
            if (gk == OP_AND) {  // set_op.rs:415-416
                float sacc = 0.0f;
                const uint32_t cnt = (uint32_t)__popc(gm);
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i)
                    if (i < cnt) sacc += pick4(val[0], val[1], val[2], val[3], b8(R.g_order4[g], i));
                gv[g] = sacc;
                if ((pm & gm) == gm) gp |= 1u << g;
            } else if (gk == OP_OR) {  // set_op.rs:169-186
                float sum = 0.0f, nd = 0.0f;
                const uint32_t ns = b8(R.g_nslots4, g);
                for (uint32_t sl = 0; sl < ns; ++sl) {
                    float m = 0.0f;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (((gm >> k) & 1u) && b8(R.g_slot4[g], k) == sl && ((pm >> k) & 1u)) m = fmaxf(m, val[k]);
                    if (m >= 0.00001f) nd += 1.0f;
                    sum += m;
                }
                gv[g] = sum * nd * nd;
                if (pm & gm) gp |= 1u << g;
            } else if (gk == OP_LEAFMAX) {  // one leaf over several lists: dedup keeps the max (search_field.rs:455-461)
                float m = 0.0f;
                bool any = false;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (((gm >> k) & 1u) && ((pm >> k) & 1u)) {
                        m = (!any || val[k] > m) ? val[k] : m;
                        any = true;
                    }
                gv[g] = m;
                if (any) gp |= 1u << g;
            } else {  // a leaf
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if ((gm >> k) & 1u) gv[g] = val[k];
                if (pm & gm) gp |= 1u << g;
            }
        }
    }
    float score;
    if (R.root_kind == OP_AND) {
        score = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i)
            if (i < R.ngroups) score += pick4(gv[0], gv[1], gv[2], gv[3], b8(R.r_order4, i));
    } else if (R.root_kind == OP_OR) {
        float sum = 0.0f, nd = 0.0f;
        for (uint32_t sl = 0; sl < R.root_nslots; ++sl) {
            float m = 0.0f;
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g)
                if (g < R.ngroups && b8(R.r_slot4, g) == sl && ((gp >> g) & 1u)) m = fmaxf(m, gv[g]);
            if (m >= 0.00001f) nd += 1.0f;
            sum += m;
        }
        score = sum * nd * nd;
    } else score = gv[0];
    // sink stages
    for (uint32_t k = 0; k < R.n_col; ++k) score = apply_col_boost(score, R.cols[k], doc);
    if (have) gb += 4u * R.n_col;
#pragma unroll
    for (uint32_t g = 0; g < 4; ++g)
        if (g < R.n_grp && (qm & b8(R.grp_mask4, g))) score *= R.grp_mult[g];
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t)
        if (t < R.n_tb && ((qm >> b8(R.tb_side4, t)) & 1u)) score *= R.tb_mult[t];
    if (R.n_loc) {  // boost.rs:11-87: 2*c*c per field with c > 1, the MINIMUM over fields (:25)
        float best = 0.0f;
        bool hav = false;
#pragma unroll
        for (uint32_t f = 0; f < 2; ++f)
            if (f < R.n_loc) {
                const uint32_t cnt = (uint32_t)__popc(pm & b8(R.loc_leaf2, f)) + (uint32_t)__popc(qm & b8(R.loc_side2, f));
                if (cnt > 1u) {
                    const float bv = 2.0f * (float)cnt * (float)cnt;
                    if (!hav || bv < best) best = bv;
                    hav = true;
                }
            }
        if (hav) score *= best;
    }
    if (have) {  // persistence.rs:164-175 count_values_for_ids: every hit counts, whatever its score
        for (uint32_t f = 0; f < R.n_facets; ++f) {
            const DFacet& fa = R.facets[f];
            if (doc >= fa.key_base && doc - fa.key_base < fa.num_keys) {
                const uint32_t row = doc - fa.key_base;
                if (fa.direct) {  // uniform: a scalar field
                    const uint32_t v = as_global(fa.direct)[row];
                    gb += 4u;
                    if (v < fa.num_values) facet_add(R.fc_keys, R.hist, fa.hist_off + v);
                    continue;
                }
                const unsigned long long e0 = as_global(fa.offsets)[row], e1 = as_global(fa.offsets)[row + 1];
                gb += 16u + 4u * (uint32_t)(e1 - e0);
                for (unsigned long long e = e0; e < e1; ++e) {
                    const uint32_t v = as_global(fa.values)[e];
                    if (v < fa.num_values) facet_add(R.fc_keys, R.hist, fa.hist_off + v);
                }
            }
        }
    }
    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
    bool pending = have && key > *cs.thr && key < cs.upper;
    while (true) {
        if (pending) {
            if (key > *cs.thr) {
                uint32_t pos = atomicAdd(cs.n, 1u);
                if (pos < cs.cap) {
                    cs.cand[pos] = key;
                    pending = false;
                }
            } else pending = false;
        }
        const int need = __syncthreads_or(pending ? 1 : 0);
        if (!need) break;
        cand_prune(cs, top_k);
    }
    if (stat) {  // uniform
        uint32_t g;
        (void)wave_excl_scan_u32(gb, &g);
        if (lane == 0) *stat += g;
    }
}

// One id list of the tile: stream its doc ids from the cursor (64 lanes x 16 B per round), set the bits of the docs inside
// [lo_bound, tile_hi) in the LDS tile, count with ballots (sorted list: the tile's entries are a prefix).  Returns the index of the
// first in-tile entry; `cur` / `nxt` advance to the first entry of the next tile.
__device__ __forceinline__ uint32_t simple_scatter_list(const uint32_t* docs, uint32_t len, uint32_t& cur, uint32_t& nxt, uint32_t* bmi, u32x4 d4, bool prefetch,
                                                        uint32_t tile_lo, uint32_t tile_hi, uint32_t lo_bound) {
    const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const uint32_t lane = threadIdx.x;
    const uint32_t c0v = cur & ~3u;
    const uint32_t nvec = (len + 3u) >> 2;
    const VQ_GLOBAL u32x4* dptr = as_global(reinterpret_cast<const u32x4*>(docs));
    uint32_t v = (c0v >> 2) + lane;
    uint32_t total_in = 0, total_lo = 0, boundary = 0xFFFFFFFFu;
    while (true) {
        const uint32_t vn = v + 64u;
        u32x4 nx = kSent;
        if (prefetch && vn < nvec) nx = dptr[vn];
        const bool ix = d4.x < tile_hi, iy = d4.y < tile_hi, iz = d4.z < tile_hi, iw = d4.w < tile_hi;
        const uint32_t mine = (uint32_t)ix + (uint32_t)iy + (uint32_t)iz + (uint32_t)iw;
        const uint32_t full = (uint32_t)__popcll(__ballot(iw));
        if (!prefetch && full == 64u && vn < nvec) nx = dptr[vn];  // sparse list: load the next round only when needed
        uint32_t n_in = full << 2;
        if (full < 64u) n_in += (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)full);
        total_in += n_in;
        const unsigned long long lom = __ballot(d4.x < lo_bound);
        if (lom) {
            const bool lx = d4.x < lo_bound, ly = d4.y < lo_bound, lz = d4.z < lo_bound, lw = d4.w < lo_bound;
            total_lo += (uint32_t)(__popcll(lom) + __popcll(__ballot(ly)) + __popcll(__ballot(lz)) + __popcll(__ballot(lw)));
            if (ix && !lx) atomicOr(&bmi[(d4.x - tile_lo) >> 5], 1u << ((d4.x - tile_lo) & 31u));
            if (iy && !ly) atomicOr(&bmi[(d4.y - tile_lo) >> 5], 1u << ((d4.y - tile_lo) & 31u));
            if (iz && !lz) atomicOr(&bmi[(d4.z - tile_lo) >> 5], 1u << ((d4.z - tile_lo) & 31u));
            if (iw && !lw) atomicOr(&bmi[(d4.w - tile_lo) >> 5], 1u << ((d4.w - tile_lo) & 31u));
        } else {
            if (ix) atomicOr(&bmi[(d4.x - tile_lo) >> 5], 1u << ((d4.x - tile_lo) & 31u));
            if (iy) atomicOr(&bmi[(d4.y - tile_lo) >> 5], 1u << ((d4.y - tile_lo) & 31u));
            if (iz) atomicOr(&bmi[(d4.z - tile_lo) >> 5], 1u << ((d4.z - tile_lo) & 31u));
            if (iw) atomicOr(&bmi[(d4.w - tile_lo) >> 5], 1u << ((d4.w - tile_lo) & 31u));
        }
        if (full < 64u) {
            const uint32_t c = mine == 0 ? d4.x : mine == 1 ? d4.y : mine == 2 ? d4.z : d4.w;
            boundary = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)full);
            break;
        }
        d4 = nx;
        v = vn;
    }
    cur = c0v + total_in;
    nxt = boundary;
    return c0v + total_lo;  // index of the first in-tile entry
}

template <uint32_t NV, bool RICH>  // NV: u32x4 bitmap vectors per lane (the tile is NV * 8192 docs); RICH: DSimple2 queries
__device__ __forceinline__ void scan_simple_body(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                 const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq, uint32_t cand_cap,
                                                 unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits, uint32_t* __restrict__ hist,
                                                 uint32_t fc_off) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t SW = kSW * NV, SWW = kSWW * NV, NW = 4u * NV;  // docs / words per tile, words per lane
    const uint32_t lane = threadIdx.x;
    uint32_t ql;
    {
        uint32_t lo = 0, hi = nq;
        const uint32_t wg = blockIdx.x;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (span_base[mid] <= wg) lo = mid;
            else hi = mid;
        }
        ql = lo;
    }
    const uint32_t span = blockIdx.x - span_base[ql];
    const uint32_t q = qmap[ql];
    const uint8_t* blob = blobs + blob_off[q];
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const uint32_t n = H->simple_n;
    const uint32_t sflags = H->simple_flags;
    const uint32_t top_k = H->top_k;
    const DList* gl = reinterpret_cast<const DList*>(blob + H->off_lists);
    const DOp* gops = reinterpret_cast<const DOp*>(blob + H->off_ops);
    const bool seq = (sflags >> 16) & 1u;
    VQ_STAMP_INIT
    VQ_STAMP_COUNT(8)
    const DSimple2* S2 = reinterpret_cast<const DSimple2*>(blob + H->off_simple2);  // RICH only
    RichShape R{};
    if constexpr (RICH) {
        R = load_rich_shape(S2, reinterpret_cast<const DColBoost*>(blob + H->off_col), H->n_col);
        R.facets = reinterpret_cast<const DFacet*>(blob + H->off_facets);
        R.n_facets = H->n_facets;
        R.hist = hist;
        if (lane < 16u) lds[kSLdsUb + lane] = __float_as_uint(S2->ub[lane]);  // (ordered in front of its first use by the barrier below)
        if (fc_off && R.n_facets) {  // uniform
            R.fc_keys = lds + fc_off;
            for (uint32_t s2 = lane; s2 < kFacetCache; s2 += 64u) {
                R.fc_keys[s2] = 0xFFFFFFFFu;
                R.fc_keys[kFacetCache + s2] = 0u;
            }
        }
    }

    SimpleLeaf lf[4];
    uint8_t order[4] = {0, 1, 2, 3}, slot[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        lf[k] = SimpleLeaf{nullptr, nullptr, nullptr, nullptr, 0u, 0.f};
        if (k < n) {
            const DList& d = gl[RICH ? (uint32_t)S2->leaf_list[k] : (uint32_t)gops[k].list_begin];
            lf[k] = SimpleLeaf{d.docs, d.scores, d.bitmap, d.rank_dir, d.len, d.term_score};
        }
    }
    uint32_t kind = OP_LEAF, nslots = 1;
    if (RICH) kind = 0xFFu;  // presence and scores follow the DSimple2 shape
    else if (n > 1) {
        kind = gops[n].kind;
        nslots = gops[n].nslots;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            order[k] = gops[n].and_order[k];
            slot[k] = gops[n].child_slot[k];
        }
    }

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* cand_n = lds + 2;
    uint32_t* qdoc = lds + kSLdsQDoc;
    uint32_t* qidx = lds + kSLdsQIdx;
    uint32_t* qmask = lds + kSLdsQMask;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + kSLdsCand);
    uint32_t* bml = lds + kSLdsCand + 2 * cand_cap;  // [number of scattered (id) lists][SWW]: lists read as bitmap images need no LDS tile
    uint32_t bslot[4];                                // LDS tile of list k
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) bslot[k] = (uint32_t)__popc(~sflags & ((1u << k) - 1u) & 0xFu) * SWW;
    CandState cs{cand, cand_n, thr, cand_cap, reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blob) + offsetof(QHeader, gthr))};
    cs.upper = H->key_upper;
    uint32_t tiles_done = 0;

    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & ~(SW - 1u));
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & ~(SW - 1u));
    const uint32_t bitmap_base = H->bitmap_base;
    const uint32_t keys_base = H->keys_base;

    // cursors of the id (scattered) lists
    uint32_t cur[4] = {0, 0, 0, 0}, nxt[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        if (k < n && !((sflags >> k) & 1u)) {
            cur[k] = wave_lower_bound(lf[k].docs, lf[k].len, span_lo);
            nxt[k] = cur[k] < lf[k].len ? as_global(lf[k].docs)[cur[k]] : 0xFFFFFFFFu;
        }
    }
    // side lists (rich queries): id lists that only the sink stages look at; each has an LDS tile behind the scattered leaves' tiles
    const uint32_t* sdocs[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t slen[4] = {0, 0, 0, 0}, scur[4] = {0, 0, 0, 0}, snxt[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, sslot[4] = {0, 0, 0, 0};
    if constexpr (RICH) {
        const uint32_t n_scatter = (uint32_t)__popc(~sflags & ((1u << n) - 1u) & 0xFu);
#pragma unroll
        for (uint32_t s = 0; s < 4; ++s)
            if (s < R.n_side) {
                const DList& d = gl[S2->side_list[s]];
                sdocs[s] = d.docs;
                slen[s] = d.len;
                sslot[s] = (n_scatter + s) * SWW;
                scur[s] = wave_lower_bound(sdocs[s], slen[s], span_lo);
                snxt[s] = scur[s] < slen[s] ? as_global(sdocs[s])[scur[s]] : 0xFFFFFFFFu;
            }
    }
    if (lane == 0) {
        *thr = 0ull;
        *cand_n = 0;
        lds[4] = 0u;
    }
    __syncthreads();
    uint32_t qlen = 0;
    unsigned long long hits = 0;
    // OR pruning state: largest posting value per operand (+inf when unknown), masks whose bound still reaches the threshold
    float or_max[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) {
            const uint16_t mr = gl[gops[k].list_begin].max_raw;
            or_max[k] = (lf[k].ts > 0.0f && mr < 0x7C00u) ? posting_value(lf[k].ts, mr) : __uint_as_float(0x7F800000u);
        }
    unsigned long long or_thr_seen = ~0ull;
    uint32_t or_live = 0xFFFFu;
    VQ_STAMP_AT(0)
    uint32_t pos = span_lo;  // sequential mode: next doc to cover
    const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const u32x4 kZero = u32x4{0u, 0u, 0u, 0u};

    while (true) {
        uint32_t head = 0xFFFFFFFFu;
        if (seq) head = pos;
        else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < n && ((sflags >> (8 + k)) & 1u)) head = nxt[k] < head ? nxt[k] : head;
        }
        if (head >= span_hi) break;
        const uint32_t tile_lo = head & ~(SW - 1u);
        const uint32_t tile_end = tile_lo + SW;
        const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;
        const uint32_t lo_bound = tile_lo > span_lo ? tile_lo : span_lo;
        pos = tile_end > tile_lo ? tile_end : 0xFFFFFFFFu;

        // issue every first load of the tile: bitmap words (+ rank directory entry) or the first id vector
        u32x4 wk[4][NV];  // lane owns the NW consecutive words [lane * NW, lane * NW + NW) of every list's tile bitmap
        uint32_t base_idx[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
#pragma unroll
            for (uint32_t h = 0; h < NV; ++h) wk[k][h] = kZero;
            if (k < n) {
                if ((sflags >> k) & 1u) {
                    const VQ_GLOBAL u32x4* gb = as_global(reinterpret_cast<const u32x4*>(lf[k].bitmap + ((tile_lo - bitmap_base) >> 5)));
#pragma unroll
                    for (uint32_t h = 0; h < NV; ++h) wk[k][h] = gb[lane * NV + h];
                    base_idx[k] = as_global(lf[k].rank_dir)[(tile_lo - bitmap_base) >> kRankShift];
                } else {
                    // a list outside the cover that fell more than a tile behind skips ahead first
                    if (nxt[k] < tile_lo && tile_lo - nxt[k] >= SW) cur[k] += wave_lower_bound(lf[k].docs + cur[k], lf[k].len - cur[k], tile_lo);
                    const uint32_t v = (cur[k] >> 2) + lane;
                    wk[k][0] = v < ((lf[k].len + 3u) >> 2) ? as_global(reinterpret_cast<const u32x4*>(lf[k].docs))[v] : kSent;
#pragma unroll
                    for (uint32_t h = 0; h < NV; ++h) reinterpret_cast<u32x4*>(bml + bslot[k])[lane * NV + h] = kZero;
                }
            }
        }
        VQ_STAMP_AT(1)
        VQ_STAMP_COUNT(7)
        // scatter the id lists (rounds of 64 x 16 B, counted with ballots; see k_tile_scan P2)
        if constexpr (RICH) {
            u32x4 sfirst[4];
#pragma unroll
            for (uint32_t s = 0; s < 4; ++s)
                if (s < R.n_side) {
                    if (snxt[s] < tile_lo && tile_lo - snxt[s] >= SW) scur[s] += wave_lower_bound(sdocs[s] + scur[s], slen[s] - scur[s], tile_lo);
                    const uint32_t v = (scur[s] >> 2) + lane;
                    sfirst[s] = v < ((slen[s] + 3u) >> 2) ? as_global(reinterpret_cast<const u32x4*>(sdocs[s]))[v] : kSent;
#pragma unroll
                    for (uint32_t h = 0; h < NV; ++h) reinterpret_cast<u32x4*>(bml + sslot[s])[lane * NV + h] = kZero;
                }
#pragma unroll
            for (uint32_t s = 0; s < 4; ++s)
                if (s < R.n_side) (void)simple_scatter_list(sdocs[s], slen[s], scur[s], snxt[s], bml + sslot[s], sfirst[s], false, tile_lo, tile_hi, lo_bound);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n && !((sflags >> k) & 1u))
                base_idx[k] = simple_scatter_list(lf[k].docs, lf[k].len, cur[k], nxt[k], bml + bslot[k], wk[k][0], (sflags >> (20 + k)) & 1u, tile_lo, tile_hi, lo_bound);
        if ((RICH ? R.prune != 0u : kind != OP_AND) && (tiles_done++ & 3u) == 0u && lane == 0) {  // (a plain AND has few survivors: nothing to prune; a rich query that prunes by bounds does, whatever its root) adopt the threshold other
                                                                          // spans of the query have published (QHeader::gthr)
            const unsigned long long g = *reinterpret_cast<volatile unsigned long long*>(cs.gthr);
            if (g > *thr) *thr = g;
        }
        __syncthreads();  // one wave: LDS atomics above are ordered before the reads below
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n && !((sflags >> k) & 1u)) {
#pragma unroll
                for (uint32_t h = 0; h < NV; ++h) wk[k][h] = reinterpret_cast<const u32x4*>(bml + bslot[k])[lane * NV + h];
            }

        VQ_STAMP_AT(2)
        // presence of the root
        u32x4 r[NV];
        uint32_t rpop = 0;
#pragma unroll
        for (uint32_t h = 0; h < NV; ++h) {
            r[h] = wk[0][h];
            if constexpr (RICH) {  // groups of leaves, then the root over the groups
                const bool root_and = R.root_kind == OP_AND;
                r[h] = root_and ? ~kZero : kZero;
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g)
                    if (g < R.ngroups) {  // uniform
                        const bool g_and = b8(R.g_kind4, g) == OP_AND;
                        const uint32_t gm = b8(R.g_mask4, g);
                        u32x4 gw = g_and ? ~kZero : kZero;
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k)
                            if ((gm >> k) & 1u) gw = g_and ? (gw & wk[k][h]) : (gw | wk[k][h]);
                        r[h] = root_and ? (r[h] & gw) : (r[h] | gw);
                    }
                if (R.has_filter) {  // uniform: the doc must be in one of the filter's lists
                    u32x4 fw = kZero;
#pragma unroll
                    for (uint32_t s2 = 0; s2 < 4; ++s2)
                        if (s2 < R.n_side && ((R.filter_mask >> s2) & 1u)) fw |= reinterpret_cast<const u32x4*>(bml + sslot[s2])[lane * NV + h];
                    r[h] &= fw;
                }
            } else if (kind == OP_AND) {
#pragma unroll
                for (uint32_t k = 1; k < 4; ++k)
                    if (k < n) r[h] &= wk[k][h];
            } else if (kind == OP_OR) {
#pragma unroll
                for (uint32_t k = 1; k < 4; ++k)
                    if (k < n) r[h] |= wk[k][h];
            }
            rpop += popc4(r[h]);
        }
        uint32_t S;
        (void)wave_excl_scan_u32(rpop, &S);
        if (S) {  // uniform
            hits += S;
            if (kind == OP_OR) {
                // Upper-bound pruning (exact): a doc present in exactly the operands of mask m scores at most
                // ub[m] = (sum over m's slots of the largest posting value of the slot's lists) * |slots|^2 — the score formula
                // evaluated on per-list maxima, and f32 add / mul are monotone.  Once the threshold exceeds ub[m] such docs are
                // counted as hits (above) but not scored.  Typical OR: after the first tiles only docs in ALL operands stay live.
                const unsigned long long thr_now = *thr;
                if (thr_now != or_thr_seen) {  // uniform, rare
                    or_thr_seen = thr_now;
                    or_live = 0xFFFFu;
                    if (thr_now != 0ull) {
                        const uint32_t tbits = (uint32_t)(thr_now >> 32);
                        or_live = 0u;
                        for (uint32_t m = 1; m < (1u << n); ++m) {
                            float sum = 0.0f, nd = 0.0f;
                            for (uint32_t sl = 0; sl < nslots; ++sl) {
                                float ms = 0.0f;
                                bool any = false;
#pragma unroll
                                for (uint32_t k = 0; k < 4; ++k)
                                    if (k < n && ((m >> k) & 1u) && slot[k] == sl) {
                                        ms = fmaxf(ms, pick4(or_max[0], or_max[1], or_max[2], or_max[3], k));
                                        any = true;
                                    }
                                if (any) nd += 1.0f;
                                sum += ms;
                            }
                            const float ub = sum * nd * nd;
                            if (!(order_f32(__float_as_uint(ub)) < tbits)) or_live |= 1u << m;  // NaN / inf bounds stay live
                        }
                    }
                }
                if (or_live != 0xFFFFu) {
                    u32x4 ev[NV];
#pragma unroll
                    for (uint32_t h = 0; h < NV; ++h) ev[h] = kZero;
                    for (uint32_t m = 1; m < (1u << n); ++m) {
                        if ((or_live >> m) & 1u) {  // uniform
#pragma unroll
                            for (uint32_t h = 0; h < NV; ++h) {
                                u32x4 t = ~kZero;
#pragma unroll
                                for (uint32_t k = 0; k < 4; ++k)
                                    if (k < n) t &= ((m >> k) & 1u) ? wk[k][h] : ~wk[k][h];
                                ev[h] |= t;
                            }
                        }
                    }
                    uint32_t epop = 0;
#pragma unroll
                    for (uint32_t h = 0; h < NV; ++h) {
                        r[h] = ev[h];
                        epop += popc4(ev[h]);
                    }
                    uint32_t S2;
                    (void)wave_excl_scan_u32(epop, &S2);
                    if (!S2) continue;  // uniform: nothing in this tile can enter the top-k
                }
            }
            VQ_STAMP_AT(3)
            // rank of each list at this lane's first word (one DPP scan per list), and — packed one byte per word — the popcounts
            // of the lane's words before word j: bytes of (pk * 0x0101..01) << 8 are the exclusive prefix sums of the bytes of pk
            uint32_t run[4] = {0, 0, 0, 0};
            unsigned long long excl[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < n) {
                    unsigned long long pk = 0ull;
                    uint32_t tot = 0;
#pragma unroll
                    for (uint32_t t = 0; t < NW; ++t) {
                        const uint32_t pc = (uint32_t)__popc(comp4(wk[k][t >> 2], t & 3u));
                        pk |= (unsigned long long)pc << (8u * t);
                        tot += pc;
                    }
                    excl[k] = (pk * 0x0101010101010101ull) << 8;
                    uint32_t dummy;
                    run[k] = base_idx[k] + wave_excl_scan_u32(tot, &dummy);
                }
            }
            VQ_STAMP_AT(4)
            // every round each lane emits its next surviving doc (lowest word, lowest bit first): the number of
            // rounds is the largest survivor count of a lane (1-2 for an AND tile, up to 128 for a dense OR tile)
            uint32_t rr[NW];
#pragma unroll
            for (uint32_t t = 0; t < NW; ++t) rr[t] = comp4(r[t >> 2], t & 3u);
            const float rich_thr_f = __uint_as_float(unorder_f32((uint32_t)(*thr >> 32)));  // the span's threshold score (NaN: none yet) — what a queued doc's bound must reach
            while (true) {  // uniform
                uint32_t j = 0u, rw = 0u;  // this lane's first word that still has a surviving doc
#pragma unroll
                for (uint32_t t = NW; t-- > 0u;)
                    if (rr[t]) {
                        j = t;
                        rw = rr[t];
                    }
                const bool has = rw != 0u;
                if (!__ballot(has)) break;
                bool keep = false;
                uint32_t qi[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, pm = 0u, m = 0u, b = 0u;
                if (has) {
                    b = (uint32_t)__ffs((int)rw) - 1u;
                    const uint32_t cleared = rw & (rw - 1u);
#pragma unroll
                    for (uint32_t t = 0; t < NW; ++t)
                        if (t == j) rr[t] = cleared;
                    const uint32_t below = (1u << b) - 1u;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        if (k < n) {
                            uint32_t word = 0u;
#pragma unroll
                            for (uint32_t t = 0; t < NW; ++t)
                                if (t == j) word = comp4(wk[k][t >> 2], t & 3u);
                            const uint32_t before = (uint32_t)(excl[k] >> (8u * j)) & 0xFFu;
                            if ((word >> b) & 1u) {
                                qi[k] = run[k] + before + (uint32_t)__popc(word & below);
                                pm |= 1u << k;
                            }
                        }
                    }
                    keep = true;
                    if constexpr (RICH) {
#pragma unroll
                        for (uint32_t s2 = 0; s2 < 4; ++s2)
                            if (s2 < R.n_side) m |= ((bml[sslot[s2] + lane * NW + j] >> b) & 1u) << s2;
                        if (R.prune) {  // uniform
                            // what this doc can score at best: the bound of its set of leaves (score tree + column boosts on list / column maxima),
                            // times the phrase / term boosts and the locality factor it really gets — the sink stages of rich_flush on the bound
                            float ub = __uint_as_float(lds[kSLdsUb + pm]);
#pragma unroll
                            for (uint32_t g = 0; g < 4; ++g)
                                if (g < R.n_grp && (m & b8(R.grp_mask4, g))) ub *= R.grp_mult[g];
#pragma unroll
                            for (uint32_t t = 0; t < 4; ++t)
                                if (t < R.n_tb && ((m >> b8(R.tb_side4, t)) & 1u)) ub *= R.tb_mult[t];
                            if (R.n_loc) {
                                float best = 0.0f;
                                bool hav = false;
#pragma unroll
                                for (uint32_t f = 0; f < 2; ++f)
                                    if (f < R.n_loc) {
                                        const uint32_t cnt = (uint32_t)__popc(pm & b8(R.loc_leaf2, f)) + (uint32_t)__popc(m & b8(R.loc_side2, f));
                                        if (cnt > 1u) {
                                            const float bv = 2.0f * (float)cnt * (float)cnt;
                                            if (!hav || bv < best) best = bv;
                                            hav = true;
                                        }
                                    }
                                if (hav) ub *= best;
                            }
                            keep = !(ub < rich_thr_f);  // (NaN while the span has no threshold: everything is kept)
                        }
                    }
                }
                const unsigned long long mask = __ballot(keep);  // the docs that are queued (the others were counted as hits above and can no longer enter the top-k)
                if (keep) {
                    const uint32_t p = qlen + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    qdoc[p] = tile_lo + ((lane * NW + j) << 5) + b;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (k < n) qidx[k * kQCap + p] = qi[k];
                    if constexpr (RICH) qmask[p] = m;
                }
                qlen += (uint32_t)__popcll(mask);
                if (qlen >= 64u) {  // uniform
                    VQ_STAMP_AT(5)
                    VQ_STAMP_COUNT(9)
                    __syncthreads();
                    uint32_t* const stat = H->stat_off ? lds + 4 : nullptr;  // (misc word 4: gathered bytes of the span, < 2^32)
                    if constexpr (RICH) rich_flush(64u, n, lf, R, qdoc, qidx, qmask, cs, top_k, stat);
                    else simple_flush(64u, n, kind, lf, order, slot, nslots, qdoc, qidx, cs, top_k, stat);
                    VQ_STAMP_AT(10)
                    // move the remainder to the front
                    const uint32_t rem = qlen - 64u;
                    uint32_t td = 0, tm = 0, ti[4] = {0, 0, 0, 0};
                    if (lane < rem) {
                        td = qdoc[64u + lane];
                        tm = qmask[64u + lane];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k)
                            if (k < n) ti[k] = qidx[k * kQCap + 64u + lane];
                    }
                    __syncthreads();
                    if (lane < rem) {
                        qdoc[lane] = td;
                        qmask[lane] = tm;
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k)
                            if (k < n) qidx[k * kQCap + lane] = ti[k];
                    }
                    __syncthreads();
                    qlen = rem;
                }
            }
            VQ_STAMP_AT(5)
        }
    }
    VQ_STAMP_AT(1)
    __syncthreads();
    if (qlen) {
        uint32_t* const stat = H->stat_off ? lds + 4 : nullptr;
        if constexpr (RICH) rich_flush(qlen, n, lf, R, qdoc, qidx, qmask, cs, top_k, stat);
        else simple_flush(qlen, n, kind, lf, order, slot, nslots, qdoc, qidx, cs, top_k, stat);
    }
    cand_prune(cs, top_k);
    {
        const uint32_t cn = *cand_n;
        unsigned long long* out = span_keys + (size_t)keys_base + (size_t)span * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cand[i] : 0ull;
    }
    if (lane == 0 && hits) atomicAdd(&num_hits[q], hits);
    if (lane == 0 && H->stat_off && lds[4]) atomicAdd(&num_hits[H->stat_off], (unsigned long long)lds[4]);
    if constexpr (RICH) {
        if (R.fc_keys) {  // uniform
            __syncthreads();
            facet_cache_flush(R.fc_keys, hist);
        }
    }
    VQ_STAMP_AT(6)
    VQ_STAMP_FLUSH
}

// 16384-doc tiles (NV = 2) cut the per-tile instruction overhead — with LDS sized by need the kernel is VALU-issue bound, not
// bandwidth bound (rocprofv3 SQ_INSTS_VALU: 60 % of the issue slots at 8192 docs per tile).  VQ_SIMPLE_NV=1 / 2 forces one width.
#ifndef VQ_RICH_WAVES
#define VQ_RICH_WAVES 4
#endif
#ifndef VQ_SIMPLE_WAVES
#define VQ_SIMPLE_WAVES 5
#endif
template <uint32_t NV, bool RICH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RICH ? VQ_RICH_WAVES : VQ_SIMPLE_WAVES, 8))) void k_scan_simple(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                    const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq,
                                                    uint32_t cand_cap, unsigned long long* __restrict__ span_keys,
                                                    unsigned long long* __restrict__ num_hits, uint32_t* __restrict__ hist, uint32_t fc_off) {
    scan_simple_body<NV, RICH>(blobs, blob_off, span_base, qmap, nq, cand_cap, span_keys, num_hits, hist, fc_off);
}

// n_scatter: LDS tiles per workgroup (scattered leaves + side lists, maximum over the launch's queries)
void launch_scan_simple(hipStream_t st, bool rich, uint32_t n_scatter, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                        const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist, bool facet_cache) {
    if (!total_spans) return;
    static const uint32_t force_nv = std::getenv("VQ_SIMPLE_NV") ? uint32_t(std::atoi(std::getenv("VQ_SIMPLE_NV"))) : 0u;
    static const bool no_fc = std::getenv("VQ_NO_FACET_CACHE") != nullptr;
    if (rich) {
        const bool fc = facet_cache && !no_fc;
        const uint32_t fc_off = fc ? kSLdsCand + 2 * cand_cap + n_scatter * kSWW * 2 : 0u;
        hipLaunchKernelGGL((k_scan_simple<2, true>), dim3(total_spans), dim3(64), scan_simple_lds_bytes(cand_cap, 2, n_scatter, fc), st, blobs, blob_off, span_base, qmap, nq,
                           cand_cap, span_keys, num_hits, hist, fc_off);
    } else if (force_nv == 1u)
        hipLaunchKernelGGL((k_scan_simple<1, false>), dim3(total_spans), dim3(64), scan_simple_lds_bytes(cand_cap, 1, n_scatter, false), st, blobs, blob_off, span_base, qmap, nq,
                           cand_cap, span_keys, num_hits, hist, 0u);
    else
        hipLaunchKernelGGL((k_scan_simple<2, false>), dim3(total_spans), dim3(64), scan_simple_lds_bytes(cand_cap, 2, n_scatter, false), st, blobs, blob_off, span_base, qmap, nq,
                           cand_cap, span_keys, num_hits, hist, 0u);
}

}  // namespace vq

// ====================================================================================================
// k_scan_wide — queries of 5..16 single-list posting leaves in a tree of depth <= 2 (DWide): the shapes of the reference's query
// generator, which ORs one leaf per term and field (src/query_generator.rs:175-246) — a flat OR over 8 leaves, an AND of two 4-leaf ORs.
// k_tile_scan interprets such trees op by op for every scored doc; here the tree shape is decoded once per wave and the tile is processed
// in two phases:
//   A  stream every leaf's words of the 8192-doc tile (dense lists: one 16 B/lane load of the bitmap image, all loads of up to 8 leaves
//      in flight together; scattered lists: the ballot-counted id scatter into LDS), fold them into the group / root presence words and
//      into a bit-sliced per-doc count of the leaves holding the doc (4 planes), keep words and per-lane ranks in LDS for phase B
//   B  only docs whose count class can still reach the top-k threshold (QHeader::prune_gbits: the score bound of a doc held by at most k
//      leaves) are queued, classes in DESCENDING order — the few docs of the highest classes raise the threshold before the many docs of
//      the low classes are looked at, so a span's first tile costs no more than any other; queued docs are scored 64 at a time, all
//      gathers of a round in flight together, the group / root arithmetic in the reference's order (set_op.rs:169-186, 393, 415-416)
// Every doc of the root words is a hit (counted), scored or not.  Same results as k_tile_scan bit for bit.
// ====================================================================================================
namespace vq {

constexpr uint32_t kWQCap = 128;  // survivor queue entries
// LDS map (u32): misc[8] | base[16] cur[16] nxt[16] gbits[16] | DWide copy [48] | per-leaf descriptors: bitmap ptr [32] rank_dir ptr [32] docs ptr [32]
//                scores ptr [32] len [16] term_score [16] | qdoc[kWQCap] | qmask[kWQCap] | cand[2*cand_cap] | qidx[L][kWQCap] | tiles[scattered leaves][kSWW]
// The tile's words of every leaf stay in REGISTERS from phase A to phase B (ML * 4 VGPRs: the kernel is instantiated for <= 8 and <= 16 leaves);
// only scattered (id) lists pass through an LDS tile.
// The tree shape and the leaves' pointers are staged once per wave: inside the tile loop nothing waits for a descriptor read from HBM
// (sub-dword fields of a descriptor in HBM cannot be scalar loads; as vector loads each one drained the bitmap loads in flight).
constexpr uint32_t kWLdsCur = 24, kWLdsNxt = 40, kWLdsGbits = 56, kWLdsW = 72, kWLdsBm = 120, kWLdsRk = 152, kWLdsDocs = 184, kWLdsSc = 216,
                   kWLdsLen = 248, kWLdsTs = 264, kWLdsQDoc = 280, kWLdsQMask = kWLdsQDoc + kWQCap, kWLdsCand = kWLdsQMask + kWQCap;
static_assert(sizeof(DWide) <= 48 * 4, "DWide copy in LDS");
size_t scan_wide_lds_bytes(uint32_t cand_cap, uint32_t n_leaves, uint32_t n_scatter) { return (size_t)(kWLdsCand + 2 * cand_cap + n_leaves * kWQCap + n_scatter * kSWW) * 4 + 16; }

// orders this wave's LDS writes / atomics before its later LDS reads (a workgroup is ONE wave: LDS executes a wave's instructions in order, so
// no barrier and — unlike __syncthreads() — no wait for the vector-memory loads in flight)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct WideShape {  // tree shape and leaf descriptors, staged in LDS
    const DWide* W;
    const unsigned long long* sc_ptr;  // [L] scores pointers
    const float* ts;                   // [L] term scores
    uint32_t L, G, f32_mask;
};

// Score queue entries [0, count) (count <= 64): gathers -> leaf values (written over the entries' qidx slots) -> groups -> root -> keys
__device__ void wide_flush(uint32_t count, const WideShape& S, const uint32_t* qdoc, const uint32_t* qmask, uint32_t* qidx, const CandState& cs, uint32_t top_k,
                           uint32_t* stat) {
    const uint32_t lane = threadIdx.x;
    const bool have = lane < count;
    const uint32_t doc = have ? qdoc[lane] : 0u;
    const uint32_t pm = have ? qmask[lane] : 0u;
    const uint32_t f32m = S.f32_mask;
    uint32_t raw[kWideMax];
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kWideMax; ++k) {
        raw[k] = 0u;
        if (k < S.L && ((pm >> k) & 1u)) {  // (k < L: uniform)
            const uint32_t idx = qidx[k * kWQCap + lane];
            const uint16_t* sp = reinterpret_cast<const uint16_t*>(S.sc_ptr[k]);
            if ((f32m >> k) & 1u) raw[k] = as_global(reinterpret_cast<const uint32_t*>(sp))[idx];  // uniform: materialised leaf
            else raw[k] = as_global(sp)[idx];
        }
    }
    if (stat) {  // uniform
        uint32_t gb = 2u * (uint32_t)__popc(pm & ~f32m) + 4u * (uint32_t)__popc(pm & f32m), g;
        (void)wave_excl_scan_u32(gb, &g);
        if (lane == 0) *stat += g;
    }
    uint32_t* const val = qidx;  // val[k * kWQCap + lane]: f32 bits of the leaf values, later of the group values at the group's first leaf
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kWideMax; ++k)
        if (k < S.L) {
            float v = 0.0f;
            if ((pm >> k) & 1u) v = ((f32m >> k) & 1u) ? __uint_as_float(raw[k]) : posting_value(S.ts[k], (uint16_t)raw[k]);
            val[k * kWQCap + lane] = __float_as_uint(v);
        }
    // groups (a lane only ever reads its own column: no synchronisation)
    const DWide* W = S.W;
    uint32_t gp = 0;
    for (uint32_t g = 0; g < S.G; ++g) {  // uniform
        const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)W->g_kind[g]), b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)W->g_begin[g]),
                       cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)W->g_count[g]);
        const uint32_t gm = ((1u << cnt) - 1u) << b0;
        if (kind == OP_LEAF) {
            if (pm & gm) gp |= 1u << g;
            continue;  // the value already sits at row b0
        }
        float gv;
        if (kind == OP_AND) {  // set_op.rs:415-416
            gv = 0.0f;
            for (uint32_t i = 0; i < cnt; ++i) gv += __uint_as_float(val[(uint32_t)W->leaf_and_order[b0 + i] * kWQCap + lane]);
            if ((pm & gm) == gm) gp |= 1u << g;
        } else {  // set_op.rs:169-186: per term slot the maximum over the present leaves, summed in slot order
            float sum = 0.0f, nd = 0.0f, m = 0.0f;
            uint32_t cur = W->leaf_slot[W->leaf_slot_order[b0]];
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t k = W->leaf_slot_order[b0 + i], sl = W->leaf_slot[k];
                if (sl != cur) {  // (the same in every lane)
                    if (m >= 0.00001f) nd += 1.0f;
                    sum += m;
                    m = 0.0f;
                    cur = sl;
                }
                if ((pm >> k) & 1u) m = fmaxf(m, __uint_as_float(val[k * kWQCap + lane]));
            }
            if (m >= 0.00001f) nd += 1.0f;
            sum += m;
            gv = sum * nd * nd;
            if (pm & gm) gp |= 1u << g;
        }
        val[b0 * kWQCap + lane] = __float_as_uint(gv);
    }
    float score;
    if (W->root_kind == OP_AND) {
        score = 0.0f;
        for (uint32_t i = 0; i < S.G; ++i) score += __uint_as_float(val[(uint32_t)W->g_begin[W->r_and_order[i]] * kWQCap + lane]);
    } else {
        float sum = 0.0f, nd = 0.0f, m = 0.0f;
        uint32_t cur = W->r_slot[W->r_slot_order[0]];
        for (uint32_t i = 0; i < S.G; ++i) {
            const uint32_t g = W->r_slot_order[i], sl = W->r_slot[g];
            if (sl != cur) {
                if (m >= 0.00001f) nd += 1.0f;
                sum += m;
                m = 0.0f;
                cur = sl;
            }
            if ((gp >> g) & 1u) m = fmaxf(m, __uint_as_float(val[(uint32_t)W->g_begin[g] * kWQCap + lane]));
        }
        if (m >= 0.00001f) nd += 1.0f;
        sum += m;
        score = sum * nd * nd;
    }
    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
    bool pending = have && key > *cs.thr && key < cs.upper;
    while (true) {
        if (pending) {
            if (key > *cs.thr) {
                const uint32_t pos = atomicAdd(cs.n, 1u);
                if (pos < cs.cap) {
                    cs.cand[pos] = key;
                    pending = false;
                }
            } else pending = false;
        }
        if (!__syncthreads_or(pending ? 1 : 0)) break;
        cand_prune(cs, top_k);
    }
}

template <uint32_t ML>  // leaves the instantiation holds in registers (8 or 16)
__device__ __forceinline__ void scan_wide_body(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off, const uint32_t* __restrict__ span_base,
                                               const uint32_t* __restrict__ qmap, uint32_t nq, uint32_t cand_cap, uint32_t ml,
                                               unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t SW = kSW, SWW = kSWW;  // 8192 docs, 256 words per tile: one u32x4 per lane and leaf
    const uint32_t lane = threadIdx.x;
    uint32_t ql;
    {
        uint32_t lo = 0, hi = nq;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (span_base[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        ql = lo;
    }
    const uint32_t span = blockIdx.x - span_base[ql];
    const uint32_t q = qmap[ql];
    const uint8_t* blob = blobs + blob_off[q];
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const uint32_t top_k = H->top_k;

    // ---- stage the tree shape and the leaves' descriptors
    DWide* const WL = reinterpret_cast<DWide*>(lds + kWLdsW);
    {
        const uint32_t* gw = reinterpret_cast<const uint32_t*>(blob + H->off_simple2);
        if (lane < (uint32_t)(sizeof(DWide) / 4)) lds[kWLdsW + lane] = gw[lane];
    }
    unsigned long long* const bm_ptr = reinterpret_cast<unsigned long long*>(lds + kWLdsBm);
    unsigned long long* const rk_ptr = reinterpret_cast<unsigned long long*>(lds + kWLdsRk);
    unsigned long long* const docs_ptr = reinterpret_cast<unsigned long long*>(lds + kWLdsDocs);
    unsigned long long* const sc_ptr = reinterpret_cast<unsigned long long*>(lds + kWLdsSc);
    uint32_t* const lens = lds + kWLdsLen;
    float* const tss = reinterpret_cast<float*>(lds + kWLdsTs);
    wave_lds_fence();
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->n_leaves), G = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->n_groups);
    if (lane < L) {
        const DList& d = reinterpret_cast<const DList*>(blob + H->off_lists)[WL->leaf_list[lane]];
        bm_ptr[lane] = (unsigned long long)(uintptr_t)d.bitmap;
        rk_ptr[lane] = (unsigned long long)(uintptr_t)d.rank_dir;
        docs_ptr[lane] = (unsigned long long)(uintptr_t)d.docs;
        sc_ptr[lane] = (unsigned long long)(uintptr_t)d.scores;
        lens[lane] = d.len;
        tss[lane] = d.term_score;
    }
    const uint32_t bitmap_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->bitmap_mask), cover_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->cover_mask),
                   prefetch_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->prefetch_mask);
    const bool seq = __builtin_amdgcn_readfirstlane((int)WL->seq) != 0;
    const bool root_and = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->root_kind) == OP_AND;
    uint32_t gfirst = 0, glast = 0, gand = 0;  // bit k: leaf k opens / closes its group, its group is an AND
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->g_begin[g]), cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->g_count[g]);
        gfirst |= 1u << b0;
        glast |= 1u << (b0 + cnt - 1u);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)WL->g_kind[g]) == OP_AND) gand |= ((1u << cnt) - 1u) << b0;
    }
    WideShape S{WL, sc_ptr, tss, L, G, (uint32_t)__builtin_amdgcn_readfirstlane((int)WL->f32_mask)};

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* cand_n = lds + 2;
    uint32_t* cur = lds + kWLdsCur;
    uint32_t* nxt = lds + kWLdsNxt;
    uint32_t* gbits = lds + kWLdsGbits;  // QHeader::prune_gbits (read once: the class loop must not wait for HBM)
    uint32_t* qdoc = lds + kWLdsQDoc;
    uint32_t* qmask = lds + kWLdsQMask;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + kWLdsCand);
    uint32_t* qidx = lds + kWLdsCand + 2 * cand_cap;  // [ml][kWQCap]
    uint32_t* tiles = qidx + ml * kWQCap;             // [scattered leaves of the query][SWW]
    CandState cs{cand, cand_n, thr, cand_cap, reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blob) + offsetof(QHeader, gthr))};
    cs.upper = H->key_upper;
    uint32_t* const stat = H->stat_off ? lds + 4 : nullptr;

    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & ~(SW - 1u));
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & ~(SW - 1u));
    const uint32_t bitmap_base = H->bitmap_base;
    const uint32_t prune_n = H->prune_n;
    if (lane < 16u) gbits[lane] = H->prune_gbits[lane];
    if (lane == 0) {
        *thr = 0ull;
        *cand_n = 0;
        lds[4] = 0u;
    }
    wave_lds_fence();
    for (uint32_t k = 0; k < L; ++k) {  // cursors of the scattered leaves
        if (!((bitmap_mask >> k) & 1u)) {
            const uint32_t* docs = reinterpret_cast<const uint32_t*>(docs_ptr[k]);
            const uint32_t len = lens[k];
            const uint32_t c = wave_lower_bound(docs, len, span_lo);
            if (lane == 0) {
                cur[k] = c;
                nxt[k] = c < len ? as_global(docs)[c] : 0xFFFFFFFFu;
            }
        }
    }
    wave_lds_fence();
    uint32_t qlen = 0, tiles_done = 0;
    unsigned long long hits = 0;
    uint32_t pos = span_lo;
    const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const u32x4 kZero = u32x4{0u, 0u, 0u, 0u};
    const uint32_t scat_mask = ~bitmap_mask & ((1u << L) - 1u);

    while (true) {
        uint32_t head = 0xFFFFFFFFu;
        if (seq) head = pos;
        else
            for (uint32_t k = 0; k < L; ++k)
                if (((cover_mask & ~bitmap_mask) >> k) & 1u) {
                    const uint32_t d = nxt[k];
                    head = d < head ? d : head;
                }
        head = (uint32_t)__builtin_amdgcn_readfirstlane((int)head);
        if (head >= span_hi) break;
        const uint32_t tile_lo = head & ~(SW - 1u);
        const uint32_t tile_end = tile_lo + SW;
        const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;
        const uint32_t lo_bound = tile_lo > span_lo ? tile_lo : span_lo;
        pos = tile_end > tile_lo ? tile_end : 0xFFFFFFFFu;
        unsigned long long gthr_now = 0ull;
        const bool adopt = (tiles_done++ & 3u) == 0u;  // now and then: adopt the threshold other spans of the query have published (QHeader::gthr);
        if (adopt && lane == 0) gthr_now = *reinterpret_cast<volatile unsigned long long*>(cs.gthr);  // (consumed behind phase A: in flight with the tile's loads)

        // ---- phase A: every first load of the tile (bitmap words + rank directory entry, or the first id vector), then the scatters
        u32x4 wv[ML];
        uint32_t rank0[ML];
#pragma unroll
        for (uint32_t k = 0; k < ML; ++k) {
            wv[k] = kZero;
            rank0[k] = 0u;
            if (k < L) {  // uniform
                if ((bitmap_mask >> k) & 1u) {
                    wv[k] = as_global(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(bm_ptr[k]) + ((tile_lo - bitmap_base) >> 5)))[lane];
                    rank0[k] = as_global(reinterpret_cast<const uint32_t*>(rk_ptr[k]))[(tile_lo - bitmap_base) >> kRankShift];
                } else {
                    const uint32_t* docs = reinterpret_cast<const uint32_t*>(docs_ptr[k]);
                    const uint32_t len = lens[k];
                    uint32_t c = cur[k];
                    const uint32_t nx = nxt[k];
                    if (nx < tile_lo && tile_lo - nx >= SW) {  // a leaf outside the cover that fell more than a tile behind skips ahead first
                        c += wave_lower_bound(docs + c, len - c, tile_lo);
                        if (lane == 0) cur[k] = c;
                    }
                    const uint32_t v = (c >> 2) + lane;
                    wv[k] = v < ((len + 3u) >> 2) ? as_global(reinterpret_cast<const u32x4*>(docs))[v] : kSent;
                    reinterpret_cast<u32x4*>(tiles + (uint32_t)__popc(scat_mask & ((1u << k) - 1u)) * SWW)[lane] = kZero;
                }
            }
        }
        if (scat_mask) {  // uniform
            wave_lds_fence();  // cursor updates and tile clears above before the scatters below
#pragma unroll
            for (uint32_t k = 0; k < ML; ++k)
                if (k < L && ((scat_mask >> k) & 1u)) {
                    uint32_t* const tile = tiles + (uint32_t)__popc(scat_mask & ((1u << k) - 1u)) * SWW;
                    uint32_t c = cur[k], nx = nxt[k];
                    rank0[k] = simple_scatter_list(reinterpret_cast<const uint32_t*>(docs_ptr[k]), lens[k], c, nx, tile, wv[k], (prefetch_mask >> k) & 1u, tile_lo, tile_hi, lo_bound);
                    if (lane == 0) {
                        cur[k] = c;
                        nxt[k] = nx;
                    }
                }
            wave_lds_fence();  // LDS atomics above before the reads below
#pragma unroll
            for (uint32_t k = 0; k < ML; ++k)
                if (k < L && ((scat_mask >> k) & 1u)) wv[k] = reinterpret_cast<const u32x4*>(tiles + (uint32_t)__popc(scat_mask & ((1u << k) - 1u)) * SWW)[lane];
        }
        // presence of the groups and of the root, bit-sliced count of the leaves holding each doc
        u32x4 root = root_and ? ~kZero : kZero, acc = kZero;
        u32x4 p0 = kZero, p1 = kZero, p2 = kZero, p3 = kZero;
#pragma unroll
        for (uint32_t k = 0; k < ML; ++k)
            if (k < L) {
                const u32x4 w = wv[k];
                if ((gfirst >> k) & 1u) acc = ((gand >> k) & 1u) ? ~kZero : kZero;
                acc = ((gand >> k) & 1u) ? (acc & w) : (acc | w);
                if ((glast >> k) & 1u) root = root_and ? (root & acc) : (root | acc);
                const u32x4 c0 = p0 & w;
                p0 ^= w;
                const u32x4 c1 = p1 & c0;
                p1 ^= c0;
                const u32x4 c2 = p2 & c1;
                p2 ^= c1;
                p3 ^= c2;
            }
        if (adopt) {  // uniform
            if (lane == 0 && gthr_now > *thr) *thr = gthr_now;
            wave_lds_fence();
        }
        uint32_t Sn;
        (void)wave_excl_scan_u32(popc4(root), &Sn);
        if (!Sn) continue;  // uniform
        hits += Sn;

        // ---- phase B: count classes in descending order (or everything at once when the query has no bound table)
        bool ranks_ready = false;  // uniform: rank0[k] += entries of leaf k in front of this lane's words — computed when the first doc is queued
        for (uint32_t cls = prune_n ? prune_n : 1u; cls >= 1u; --cls) {  // uniform
            u32x4 sel = root;
            if (prune_n) {
                const uint32_t thr_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(*thr >> 32));
                if (thr_hi && (uint32_t)__builtin_amdgcn_readfirstlane((int)gbits[cls]) < thr_hi) break;  // neither this class nor a lower one can reach the threshold
                sel &= ((cls & 1u) ? p0 : ~p0) & ((cls & 2u) ? p1 : ~p1) & ((cls & 4u) ? p2 : ~p2) & ((cls & 8u) ? p3 : ~p3);
            }
            uint32_t rr[4] = {sel.x, sel.y, sel.z, sel.w};
            while (true) {  // uniform: every round each lane queues its next doc of the class
                uint32_t j = 0u, rw = 0u;
#pragma unroll
                for (uint32_t t = 4; t-- > 0u;)
                    if (rr[t]) {
                        j = t;
                        rw = rr[t];
                    }
                const bool has = rw != 0u;
                const unsigned long long mask = __ballot(has);
                if (!mask) break;
                if (!ranks_ready) {  // uniform
                    ranks_ready = true;
#pragma unroll
                    for (uint32_t k = 0; k < ML; ++k)
                        if (k < L) {
                            uint32_t dummy;
                            rank0[k] += wave_excl_scan_u32(popc4(wv[k]), &dummy);
                        }
                }
                if (has) {
                    const uint32_t b = (uint32_t)__ffs((int)rw) - 1u;
                    const uint32_t cleared = rw & (rw - 1u);
#pragma unroll
                    for (uint32_t t = 0; t < 4; ++t)
                        if (t == j) rr[t] = cleared;
                    const uint32_t below = (1u << b) - 1u;
                    const uint32_t p = qlen + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    qdoc[p] = tile_lo + ((lane * 4u + j) << 5) + b;
                    uint32_t m16 = 0;
#pragma unroll
                    for (uint32_t k = 0; k < ML; ++k)
                        if (k < L) {
                            const u32x4 w4 = wv[k];
                            const uint32_t word = comp4(w4, j);
                            if ((word >> b) & 1u) {
                                m16 |= 1u << k;
                                const uint32_t before = (j > 0u ? (uint32_t)__popc(w4.x) : 0u) + (j > 1u ? (uint32_t)__popc(w4.y) : 0u) + (j > 2u ? (uint32_t)__popc(w4.z) : 0u);
                                qidx[k * kWQCap + p] = rank0[k] + before + (uint32_t)__popc(word & below);
                            }
                        }
                    qmask[p] = m16;
                }
                qlen += (uint32_t)__popcll(mask);
                if (qlen >= 64u) {  // uniform
                    wave_lds_fence();
                    wide_flush(64u, S, qdoc, qmask, qidx, cs, top_k, stat);
                    const uint32_t rem = qlen - 64u;  // move the remainder to the front (source and destination are disjoint)
                    wave_lds_fence();
                    if (lane < rem) {
                        const uint32_t td = qdoc[64u + lane], tm = qmask[64u + lane];
                        for (uint32_t k = 0; k < L; ++k) qidx[k * kWQCap + lane] = qidx[k * kWQCap + 64u + lane];
                        qdoc[lane] = td;
                        qmask[lane] = tm;
                    }
                    wave_lds_fence();
                    qlen = rem;
                }
            }
            if (!prune_n) break;
        }
    }
    wave_lds_fence();
    if (qlen) wide_flush(qlen, S, qdoc, qmask, qidx, cs, top_k, stat);
    cand_prune(cs, top_k);
    {
        const uint32_t cn = *cand_n;
        unsigned long long* out = span_keys + (size_t)H->keys_base + (size_t)span * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cand[i] : 0ull;
    }
    if (lane == 0 && hits) atomicAdd(&num_hits[q], hits);
    if (lane == 0 && H->stat_off && lds[4]) atomicAdd(&num_hits[H->stat_off], (unsigned long long)lds[4]);
}

template <uint32_t ML>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ML <= 8 ? 4 : 3, 8))) void k_scan_wide(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                   const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq, uint32_t cand_cap,
                                                   uint32_t ml, unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits) {
    scan_wide_body<ML>(blobs, blob_off, span_base, qmap, nq, cand_cap, ml, span_keys, num_hits);
}

void launch_scan_wide(hipStream_t st, uint32_t max_leaves, uint32_t max_scatter, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                      const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits) {
    if (!total_spans) return;
    const size_t lds_bytes = scan_wide_lds_bytes(cand_cap, max_leaves, max_scatter);
    if (max_leaves <= 8)
        hipLaunchKernelGGL((k_scan_wide<8>), dim3(total_spans), dim3(64), lds_bytes, st, blobs, blob_off, span_base, qmap, nq, cand_cap, max_leaves, span_keys, num_hits);
    else
        hipLaunchKernelGGL((k_scan_wide<16>), dim3(total_spans), dim3(64), lds_bytes, st, blobs, blob_off, span_base, qmap, nq, cand_cap, max_leaves, span_keys, num_hits);
}

}  // namespace vq

// ====================================================================================================
// k_dict_scan (K9) — fuzzy / prefix term expansion: one lane per dictionary term, Myers / Hyyrö bit-vector
// edit distance of the query (pattern, <= 64 code points) against the term (text).  Decides exactly what the
// reference's Levenshtein DFA accepts (search_field.rs:85-95): distance(term, query) <= max_d, with adjacent
// transpositions at cost one when requested, or — for starts_with — the minimum over all prefixes of the term.
// Code points are compared as stored (the host hands over the lower-cased image for case-insensitive scans).
// ====================================================================================================
namespace vq {

// Myers / Hyyrö bit-vector recurrence over the term's code points, in 32-bit words when the pattern has <= 32 code points (half the VALU
// work).  Whole-term matching leaves the loop as soon as the distance can no longer come back under max_d (the last row of the DP table falls
// by at most one per text character): most terms of a dictionary are out after two or three characters.
// (`text` is either the LDS stage or the HBM image: the two callers are separate instantiations on purpose — a pointer selected at run time
// between the two becomes a FLAT load, which costs several hundred cycles per character of every surviving pair)
template <class Word>
__device__ __forceinline__ bool dict_match(uint32_t m, uint32_t max_d, bool transposition, bool prefix, uint32_t n, const uint16_t* text,
                                           const unsigned long long* peq_p, const uint16_t* q) {
    const Word one = 1;
    const Word top = one << (m - 1);
    Word Pv = m == sizeof(Word) * 8 ? ~Word(0) : ((one << m) - one), Mv = 0, prevEq = 0, prevD0 = ~Word(0);
    uint32_t score = m;
    uint32_t best = m;  // distance of the empty prefix
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t c = text[i];
        Word Eq;
        if (c < 128u) Eq = (Word)peq_p[c];
        else {
            Eq = 0;
            for (uint32_t j = 0; j < m; ++j) Eq |= (Word)(q[j] == c) << j;
        }
        Word D0 = (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv;
        if (transposition) D0 |= (((~prevD0) & Eq) << 1) & prevEq;  // Hyyrö 2003: adjacent transposition, cost one
        Word Ph = Mv | ~(D0 | Pv);
        Word Mh = Pv & D0;
        if (Ph & top) ++score;
        else if (Mh & top) --score;
        Ph = (Ph << 1) | one;
        Mh <<= 1;
        Pv = Mh | ~(D0 | Ph);
        Mv = Ph & D0;
        prevEq = Eq;
        prevD0 = D0;
        best = score < best ? score : best;
        if (!prefix && score > max_d + (n - 1u - i)) return false;  // cannot come back under max_d
    }
    return (prefix ? best : score) <= max_d;
}

// Distance the reference SCORES a hit with (search_field.rs:691-732): full bit-vector recurrence of the lower-cased term (pattern, <= 64 code
// points) over the lower-cased hit; TRANS: adjacent transpositions cost one (the scoring automaton), else plain Levenshtein (its fallback).
// (peq_q: the pattern's match-mask table for code points < 128, or null)
template <bool TRANS>
__device__ __forceinline__ uint32_t dict_full_distance(const uint16_t* q, const unsigned long long* peq_q, uint32_t m, const uint16_t* __restrict__ text, uint32_t n) {
    if (m == 0) return n;
    const unsigned long long top = 1ull << (m - 1);
    unsigned long long Pv = m == 64 ? ~0ull : ((1ull << m) - 1ull), Mv = 0ull, prevEq = 0ull, prevD0 = ~0ull;
    uint32_t score = m;
    for (uint32_t i = 0; i < n; ++i) {
        const uint16_t c = text[i];
        unsigned long long Eq;
        if (peq_q && c < 128u) Eq = peq_q[c];
        else {
            Eq = 0ull;
            for (uint32_t j = 0; j < m; ++j) Eq |= (unsigned long long)(q[j] == c) << j;
        }
        unsigned long long D0 = (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv;
        if (TRANS) D0 |= (((~prevD0) & Eq) << 1) & prevEq;
        unsigned long long Ph = Mv | ~(D0 | Pv);
        unsigned long long Mh = Pv & D0;
        if (Ph & top) ++score;
        else if (Mh & top) --score;
        Ph = (Ph << 1) | 1ull;
        Mh <<= 1;
        Pv = Mh | ~(D0 | Ph);
        Mv = Ph & D0;
        prevEq = Eq;
        prevD0 = D0;
    }
    return score;
}

constexpr uint32_t kDictGroup = 16;    // probes one block answers per pass over its terms
constexpr uint32_t kDictStage = 4096;  // code points of a round's 256 terms staged in LDS (longer stretches are read from HBM)
constexpr uint32_t kDictRounds = 8;    // rounds of 256 consecutive terms per block: the probes' match-mask tables are built once for all of them

// One block = kDictRounds rounds of 256 consecutive dictionary terms (a round's code points are one contiguous stretch of the CSR image: staged into LDS with coalesced
// loads) x a group of kDictGroup probes.  Per probe a match-mask table Peq[c] (bit j: query[j] == c) for c < 128 lives in LDS, so the per-character
// step of the bit-vector recurrence is one LDS read instead of an m-step compare loop; other code points take the compare loop.
// All probes of a launch scan the same dictionary image (the host groups them).
// The recurrence for a term of <= kDictShort ASCII code points staged in LDS: all its characters, then all their match masks, are read in two
// batches of independent LDS loads — the character-by-character loop of dict_match pays two dependent LDS round trips per character.
constexpr uint32_t kDictShort = 12;
template <class Word>
__device__ __forceinline__ bool dict_match_short(uint32_t m, uint32_t max_d, bool transposition, bool prefix, uint32_t n, const uint16_t* text,
                                                 const unsigned long long* peq_p, bool* ascii) {
    uint32_t c[kDictShort];
    bool all_ascii = true;
#pragma unroll
    for (uint32_t i = 0; i < kDictShort; ++i) {
        c[i] = i < n ? (uint32_t)text[i] : 0u;
        all_ascii = all_ascii && c[i] < 128u;
    }
    *ascii = all_ascii;
    if (!all_ascii) return false;
    Word eq[kDictShort];
#pragma unroll
    for (uint32_t i = 0; i < kDictShort; ++i) eq[i] = (Word)peq_p[c[i]];
    const Word one = 1;
    const Word top = one << (m - 1);
    Word Pv = m == sizeof(Word) * 8 ? ~Word(0) : ((one << m) - one), Mv = 0, prevEq = 0, prevD0 = ~Word(0);
    uint32_t score = m, best = m;
    bool dead = false;
#pragma unroll
    for (uint32_t i = 0; i < kDictShort; ++i) {
        if (i < n && !dead) {
            const Word Eq = eq[i];
            Word D0 = (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv;
            if (transposition) D0 |= (((~prevD0) & Eq) << 1) & prevEq;
            Word Ph = Mv | ~(D0 | Pv);
            Word Mh = Pv & D0;
            if (Ph & top) ++score;
            else if (Mh & top) --score;
            Ph = (Ph << 1) | one;
            Mh <<= 1;
            Pv = Mh | ~(D0 | Ph);
            Mv = Ph & D0;
            prevEq = Eq;
            prevD0 = D0;
            best = score < best ? score : best;
            if (!prefix && score > max_d + (n - 1u - i)) dead = true;
        }
    }
    return !dead && (prefix ? best : score) <= max_d;
}

__global__ __launch_bounds__(256) void k_dict_scan(const DictProbe* __restrict__ probes, uint32_t probe_base, uint32_t n_probes, const uint32_t* __restrict__ off,
                                                   const uint16_t* __restrict__ chars, const uint16_t* __restrict__ low_chars, uint32_t num_terms,
                                                   uint32_t* __restrict__ out_count, uint32_t out_cap, DictMatch* __restrict__ out) {
    __shared__ unsigned long long peq[kDictGroup][128];
    __shared__ uint16_t stage[kDictStage];
    __shared__ uint16_t qch[kDictGroup][64];
    __shared__ uint32_t pm[kDictGroup], pmaxd[kDictGroup], pflags[kDictGroup];
    __shared__ unsigned long long psig[kDictGroup];
    __shared__ uint32_t soff[257];                   // a round's term offsets, relative to its first code point
    __shared__ uint16_t queue[256 * kDictGroup];     // (term in round | probe << 8) pairs that passed the filters
    __shared__ uint32_t qn;
    VQ_STAMP_INIT
    VQ_STAMP_COUNT(8)
    const uint32_t tid = threadIdx.x;
    const uint32_t p0 = blockIdx.y * kDictGroup;
    const uint32_t np = n_probes - p0 < kDictGroup ? n_probes - p0 : kDictGroup;
    for (uint32_t x = tid; x < np * 64u; x += 256u) qch[x >> 6][x & 63u] = probes[p0 + (x >> 6)].query[x & 63u];
    if (tid < kDictGroup) {
        pm[tid] = tid < np ? probes[p0 + tid].m : 0u;
        pmaxd[tid] = tid < np ? probes[p0 + tid].max_d : 0u;
        pflags[tid] = tid < np ? probes[p0 + tid].flags : 0u;
    }
    for (uint32_t x = tid; x < kDictGroup * 128u; x += 256u) peq[x >> 7][x & 127u] = 0ull;
    __syncthreads();
    // the probes' tables, built once per block and used for all its kDictRounds x 256 terms: Peq[c] bit j = (query[j] == c), one LDS atomic
    // per character of a query
    for (uint32_t x = tid; x < np * 64u; x += 256u) {
        const uint32_t p = x >> 6, j = x & 63u;
        if (j < pm[p] && qch[p][j] < 128u) atomicOr(&peq[p][qch[p][j]], 1ull << j);
    }
    // Character-set filter in front of the recurrence: every edit (insert, delete, substitute; a transposition none) takes at most ONE distinct
    // character of the query out of the term, so a term within max_d of the query — or with a prefix that is — lacks at most max_d of the
    // query's distinct characters.  Characters are hashed into a 64-bit signature (collisions only hide a lacking character: the count is a
    // lower bound, no match is lost); a random dictionary term fails this test on a few register operations, without touching LDS.
    if (tid < kDictGroup) {
        unsigned long long sq = 0ull;
        for (uint32_t j = 0; j < pm[tid]; ++j) sq |= 1ull << (((uint32_t)qch[tid][j] * 2654435761u) >> 26);
        psig[tid] = sq;
    }
    __syncthreads();
    // the group's parameters in scalar registers: the filter loop below touches no memory at all
    uint32_t pk[kDictGroup], sg_lo[kDictGroup], sg_hi[kDictGroup];  // pk: m | max_d << 8 | flags << 16 | valid << 31
#pragma unroll
    for (uint32_t p = 0; p < kDictGroup; ++p) {
        pk[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pm[p] | (pmaxd[p] << 8) | (pflags[p] << 16) | (p < np ? 0x80000000u : 0u)));
        sg_lo[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)psig[p]);
        sg_hi[p] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(psig[p] >> 32));
    }
    // A round's inputs — 257 offsets and up to kDictStage code points — are loaded into registers one round AHEAD: the block's rounds are
    // consecutive stretches of the image, so the next stretch starts where this one ends, and its loads are in flight while this round's
    // pairs are filtered and matched.
    const uint32_t total_chars = off[num_terms];
    const uint32_t first_t0 = blockIdx.x * kDictRounds * 256u;
    uint32_t n_off0 = 0, n_off1 = 0;
    uint16_t n_ch[kDictStage / 256];
    auto prefetch = [&](uint32_t t0, uint32_t base) {
        if (t0 >= num_terms) return;  // uniform
        const uint32_t last = num_terms;  // off[] has num_terms + 1 entries
        n_off0 = off[t0 + tid <= last ? t0 + tid : last];
        if (tid == 0) n_off1 = off[t0 + 256u <= last ? t0 + 256u : last];
#pragma unroll
        for (uint32_t k = 0; k < kDictStage / 256u; ++k) {
            const uint32_t at = base + k * 256u + tid;
            n_ch[k] = at < total_chars ? chars[at] : (uint16_t)0;
        }
    };
    uint32_t next_base = first_t0 < num_terms ? off[first_t0] : 0u;
    prefetch(first_t0, next_base);
    VQ_STAMP_AT(0)
    for (uint32_t round = 0; round < kDictRounds; ++round) {
        const uint32_t t0 = first_t0 + round * 256u;
        if (t0 >= num_terms) break;  // uniform
        const uint32_t t_end = t0 + 256u < num_terms ? t0 + 256u : num_terms;
        const uint32_t base = next_base;
        __syncthreads();  // the previous round's readers of `stage` / `soff` / `queue` are done
        soff[tid] = n_off0 - base;
        if (tid == 0) {
            soff[256] = n_off1 - base;
            qn = 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kDictStage / 256u; ++k) stage[k * 256u + tid] = n_ch[k];
        __syncthreads();
        const uint32_t stretch = soff[t_end - t0];
        const uint32_t staged = stretch < kDictStage ? stretch : kDictStage;
        next_base = base + stretch;
        prefetch(t0 + 256u, next_base);
        VQ_STAMP_AT(1)
        VQ_STAMP_COUNT(7)
        // phase A — one lane per term: the cheap filters against every probe of the group (registers only); the survivors (about 2 % of the
        // pairs) of a wave are queued with one reservation
        uint32_t pmask = 0u;  // bit p: this lane's term survived against probe p
        if (t0 + tid < t_end) {
            const uint32_t b = soff[tid], n = soff[tid + 1] - b;
            const bool in_stage = b + n <= staged;
            unsigned long long sig_t = 0ull;
            if (in_stage)
                for (uint32_t i = 0; i < n; ++i) sig_t |= 1ull << (((uint32_t)stage[b + i] * 2654435761u) >> 26);
            else
                for (uint32_t i = 0; i < n; ++i) sig_t |= 1ull << (((uint32_t)chars[base + b + i] * 2654435761u) >> 26);
            const uint32_t nt_lo = ~(uint32_t)sig_t, nt_hi = ~(uint32_t)(sig_t >> 32);
#pragma unroll
            for (uint32_t p = 0; p < kDictGroup; ++p) {
                const uint32_t m = pk[p] & 0xFFu, max_d = (pk[p] >> 8) & 0xFFu;
                const bool prefix = (pk[p] >> 17) & 1u, valid = pk[p] >> 31;
                const bool len_ok = prefix || !(n > m + max_d || n + max_d < m);
                const uint32_t lacking = (uint32_t)__popc(sg_lo[p] & nt_lo) + (uint32_t)__popc(sg_hi[p] & nt_hi);
                pmask |= (valid && len_ok && lacking <= max_d) ? (1u << p) : 0u;
            }
        }
        {
            uint32_t total;
            const uint32_t excl = wave_excl_scan_u32((uint32_t)__popc(pmask), &total);
            uint32_t wbase = 0u;
            if (total) {  // uniform
                if ((tid & 63u) == 0u) wbase = atomicAdd(&qn, total);
                wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            }
            uint32_t w = wbase + excl;
            while (pmask) {
                const uint32_t p = (uint32_t)__ffs(pmask) - 1u;
                pmask &= pmask - 1u;
                queue[w++] = (uint16_t)(tid | (p << 8));
            }
        }
        __syncthreads();
        VQ_STAMP_AT(2)
        // phase B — one lane per surviving (term, probe) pair: the recurrence runs with every lane busy instead of once per probe for a wave in
        // which one lane survived
        const uint32_t pairs = qn;
#ifdef VQ_STAMP
        _acc[11] += pairs;
#endif
        for (uint32_t x = tid; x < pairs; x += 256u) {
            const uint32_t lt = queue[x] & 255u, p = queue[x] >> 8;
            const uint32_t b = soff[lt], n = soff[lt + 1] - b;
            const bool in_stage = b + n <= staged;
            const uint32_t m = pm[p], max_d = pmaxd[p];
            const bool transposition = pflags[p] & 1u, prefix = pflags[p] & 2u;
            bool match, done = false;
            if (m == 0) {
                match = prefix || n <= max_d;
                done = true;
            } else if (in_stage && n <= kDictShort) {
                if (m <= 32u) match = dict_match_short<uint32_t>(m, max_d, transposition, prefix, n, &stage[b], peq[p], &done);
                else match = dict_match_short<unsigned long long>(m, max_d, transposition, prefix, n, &stage[b], peq[p], &done);
            }
            if (!done) {
                if (in_stage) {
                    if (m <= 32u) match = dict_match<uint32_t>(m, max_d, transposition, prefix, n, &stage[b], peq[p], qch[p]);
                    else match = dict_match<unsigned long long>(m, max_d, transposition, prefix, n, &stage[b], peq[p], qch[p]);
                } else {
                    if (m <= 32u) match = dict_match<uint32_t>(m, max_d, transposition, prefix, n, chars + base + b, peq[p], qch[p]);
                    else match = dict_match<unsigned long long>(m, max_d, transposition, prefix, n, chars + base + b, peq[p], qch[p]);
                }
            }
            if (match) {  // rare: what the hit's score needs is computed here, on the lower-cased image
                uint32_t info = 0;
                if ((pflags[p] & 4u) && in_stage) {  // scored with the string and over the image it was matched with: everything is in LDS already
                    const uint16_t* text = &stage[b];
                    const uint32_t osa = dict_full_distance<true>(qch[p], peq[p], m, text, n), lev = dict_full_distance<false>(qch[p], peq[p], m, text, n);
                    bool starts = n >= m;
                    for (uint32_t i = 0; starts && i < m; ++i) starts = text[i] == qch[p][i];
                    info = (osa < 255u ? osa : 255u) | ((lev < 255u ? lev : 255u) << 8) | ((starts ? 1u : 0u) << 16);
                } else {
                    const DictProbe& P = probes[p0 + p];
                    const uint32_t lm = P.lm;
                    if (lm != 0xFFFFFFFFu) {
                        const uint16_t* text = low_chars + base + b;
                        const uint32_t osa = dict_full_distance<true>(P.lquery, nullptr, lm, text, n), lev = dict_full_distance<false>(P.lquery, nullptr, lm, text, n);
                        bool starts = n >= lm;
                        for (uint32_t i = 0; starts && i < lm; ++i) starts = text[i] == P.lquery[i];
                        info = (osa < 255u ? osa : 255u) | ((lev < 255u ? lev : 255u) << 8) | ((starts ? 1u : 0u) << 16);
                    }
                }
                const uint32_t pos = atomicAdd(out_count, 1u);
                if (pos < out_cap) out[pos] = DictMatch{probe_base + p0 + p, t0 + lt, info};
            }
        }
        VQ_STAMP_AT(3)
    }
    VQ_STAMP_FLUSH
}

void launch_dict_scan(hipStream_t st, const DictProbe* d_probes, uint32_t probe_base, uint32_t n_probes, const uint32_t* off, const uint16_t* chars, const uint16_t* low_chars,
                      uint32_t num_terms, uint32_t* out_count, uint32_t out_cap, DictMatch* out) {
    if (!n_probes || !num_terms) return;
    hipLaunchKernelGGL(k_dict_scan, dim3((num_terms + 256u * kDictRounds - 1u) / (256u * kDictRounds), (n_probes + kDictGroup - 1u) / kDictGroup), dim3(256), 0, st, d_probes, probe_base, n_probes, off, chars,
                       low_chars, num_terms, out_count, out_cap, out);
}

}  // namespace vq

// ====================================================================================================
// Text locality of a field whose text ids are not anchors (K7, boost.rs:34-87), as a pre-pass over the batch's (request, field) jobs:
//   k_loc_gather   the token -> text rows of every query term, copied into one buffer (a job = one contiguous slice)
//   [segmented radix sort of the text ids]
//   k_loc_expand   a text id occurring c > 1 times in its job's slice (c counts list entries, boost.rs:51-56) is expanded through
//                  text_id_to_anchor: one (anchor << 32 | f32 bits of 2*c*c) pair per anchor (count pass, then write pass)
//   [segmented radix sort of the pairs: by anchor, then by boost]
//   k_loc_compact  the first pair of every anchor — its smallest boost, which is what the reference's reversed max_by keeps (boost.rs:25) —
//                  goes to the job's (doc, f32) list, padded like a materialised leaf; the scan kernels read it as a LIST_F32 id list
// ====================================================================================================
namespace vq {

__global__ __launch_bounds__(256) void k_loc_gather(const LocRow* __restrict__ rows, const uint32_t* __restrict__ table, uint32_t* __restrict__ gathered) {
    const LocRow r = rows[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < r.len; i += 256u) gathered[r.dst + i] = table[r.src + i];
}
void launch_loc_gather(hipStream_t st, const LocRow* rows, uint32_t n_rows, const uint32_t* table, uint32_t* gathered) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_loc_gather, dim3(n_rows), dim3(256), 0, st, rows, table, gathered);
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_loc_expand(const LocJob* __restrict__ jobs, uint32_t n_jobs, const uint32_t* __restrict__ ids, uint32_t n,
                                                    uint32_t* __restrict__ counters, unsigned long long* __restrict__ pairs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = n_jobs;  // the job whose slice holds element i (slices are consecutive, in job order)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].seg_begin <= i) lo = mid;
        else hi = mid;
    }
    const LocJob J = jobs[lo];
    if (i >= J.seg_end) return;
    const uint32_t t = ids[i];
    if (i + 1 < J.seg_end && ids[i + 1] == t) return;  // not the last entry of its run
    uint32_t a = J.seg_begin, b = i;                    // first entry of the run: lower bound of t inside [seg_begin, i]
    while (a < b) {
        const uint32_t mid = a + ((b - a) >> 1);
        if (ids[mid] < t) a = mid + 1;
        else b = mid;
    }
    const uint32_t c = i - a + 1u;
    if (c <= 1u) return;
    if (t < J.t2a_key_base || t - J.t2a_key_base >= J.t2a_num_keys) return;
    const uint32_t row = t - J.t2a_key_base;
    const uint32_t len = J.t2a_len[row];
    if (!len) return;
    const uint32_t at = atomicAdd(&counters[lo], len);
    if (WRITE) {
        const float boost = 2.0f * (float)c * (float)c;  // boost.rs:70,80
        const unsigned long long low = (unsigned long long)__float_as_uint(boost);
        const uint32_t* anchors = J.t2a_vals + J.t2a_start[row];
        for (uint32_t e = 0; e < len; ++e) pairs[(size_t)J.pair_begin + at + e] = ((unsigned long long)anchors[e] << 32) | low;
    }
}
void launch_loc_expand(hipStream_t st, bool write, const LocJob* jobs, uint32_t n_jobs, const uint32_t* sorted_text_ids, uint32_t n, uint32_t* counters,
                       unsigned long long* pairs) {
    if (!n || !n_jobs) return;
    if (write) hipLaunchKernelGGL(k_loc_expand<true>, dim3((n + 255u) / 256u), dim3(256), 0, st, jobs, n_jobs, sorted_text_ids, n, counters, pairs);
    else hipLaunchKernelGGL(k_loc_expand<false>, dim3((n + 255u) / 256u), dim3(256), 0, st, jobs, n_jobs, sorted_text_ids, n, counters, pairs);
}

// one wave per job: an ordered compaction of its sorted pairs
// K10 — one wave per 1:n boost list: its sorted value ids, 4 x 64 at a time (all lookups of a round in flight together), keep the boosted
// ones (boost_valid_to_value has an entry), look the anchor up (value_id_to_anchor, first value of the row: boost.rs:455-463), and compact
// the (anchor, value) pairs of the shard's docs into the job's padded (doc, f32) list in value-id order.  Along the way: are the anchors
// non-decreasing, does one repeat (the compiler needs both, see Boost1nJob).
__global__ __launch_bounds__(64) void k_b1n_map(const B1nJob* __restrict__ jobs, const uint32_t* __restrict__ vids, uint32_t* __restrict__ out_docs,
                                                float* __restrict__ out_vals, B1nResult* __restrict__ results) {
    const B1nJob J = jobs[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    uint32_t written = 0, total = 0, flags = 0;
    uint32_t prev = 0;       // anchor of the last pair seen so far (any shard)
    bool have_prev = false;  // uniform
    for (uint32_t base = J.seg_begin; base < J.seg_end; base += 256u) {  // uniform
        uint32_t anchor[4];
        float value[4];
        bool keep[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t i = base + u * 64u + lane;
            keep[u] = false;
            anchor[u] = 0u;
            value[u] = 0.0f;
            if (i < J.seg_end) {
                const uint32_t v = vids[i];
                if (v >= J.boost_key_base && v - J.boost_key_base < J.boost_num_keys && v >= J.to_anchor_key_base && v - J.to_anchor_key_base < J.to_anchor_num_keys) {
                    const uint32_t rb = v - J.boost_key_base, ra = v - J.to_anchor_key_base;
                    const bool present = !J.boost_present || ((J.boost_present[rb >> 5] >> (rb & 31u)) & 1u);
                    const unsigned long long a0 = J.to_anchor_off[ra], a1 = J.to_anchor_off[ra + 1];
                    if (present && a1 > a0) {
                        keep[u] = true;
                        anchor[u] = J.to_anchor_vals[a0];
                        value[u] = J.boost_values[rb];
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const unsigned long long m = __ballot(keep[u]);
            if (!m) continue;  // uniform
            // order checks against the pair in front (previous kept lane, or the last pair of the rounds before)
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            const uint32_t got = (uint32_t)__shfl((int)anchor[u], below ? 63 - (int)__clzll((long long)below) : (int)lane);  // (executed by every lane)
            const uint32_t before = below ? got : prev;
            const bool has_before = below ? true : have_prev;
            if (keep[u] && has_before) {
                if (anchor[u] < before) flags |= 1u;
                if (anchor[u] == before) flags |= 2u;
            }
            const bool in_shard = keep[u] && anchor[u] >= J.doc_lo && anchor[u] < J.doc_hi;
            const unsigned long long ms = __ballot(in_shard);
            if (in_shard) {
                const uint32_t pos = J.out_off + written + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
                out_docs[pos] = anchor[u];
                out_vals[pos] = value[u];
            }
            written += (uint32_t)__popcll(ms);
            total += (uint32_t)__popcll(m);
            prev = (uint32_t)__shfl((int)anchor[u], 63 - (int)__clzll((long long)m));
            have_prev = true;
        }
    }
    flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__ballot(flags & 1u) ? 1u : 0u)) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(__ballot(flags & 2u) ? 2u : 0u));
    if (lane < 8u) {  // sentinels behind the list: the scans read whole 16-byte vectors
        out_docs[J.out_off + written + lane] = 0xFFFFFFFFu;
        out_vals[J.out_off + written + lane] = 0.0f;
    }
    if (lane == 0) results[blockIdx.x] = B1nResult{written, total, flags, 0u};
}
void launch_b1n_map(hipStream_t st, const B1nJob* jobs, uint32_t n_jobs, const uint32_t* sorted_value_ids, uint32_t* out_docs, float* out_vals, B1nResult* results) {
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_b1n_map, dim3(n_jobs), dim3(64), 0, st, jobs, sorted_value_ids, out_docs, out_vals, results);
}

__global__ __launch_bounds__(64) void k_loc_compact(const LocJob* __restrict__ jobs, const unsigned long long* __restrict__ sorted, uint32_t* __restrict__ out_docs,
                                                    float* __restrict__ out_vals, uint32_t* __restrict__ out_len) {
    const LocJob J = jobs[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    uint32_t written = 0;
    for (uint32_t base = J.pair_begin; base < J.pair_end; base += 64u) {  // uniform
        const uint32_t i = base + lane;
        bool keep = false;
        unsigned long long p = 0ull;
        if (i < J.pair_end) {
            p = sorted[i];
            keep = i == J.pair_begin || (uint32_t)(sorted[i - 1] >> 32) != (uint32_t)(p >> 32);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const uint32_t pos = J.out_off + written + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            out_docs[pos] = (uint32_t)(p >> 32);
            out_vals[pos] = __uint_as_float((uint32_t)p);
        }
        written += (uint32_t)__popcll(m);
    }
    if (lane < 8u) {  // list padding: sentinel docs up to the vector width and beyond
        out_docs[J.out_off + written + lane] = 0xFFFFFFFFu;
        out_vals[J.out_off + written + lane] = 0.0f;
    }
    if (lane == 0) out_len[blockIdx.x] = written;
}
void launch_loc_compact(hipStream_t st, const LocJob* jobs, uint32_t n_jobs, const unsigned long long* sorted_pairs, uint32_t* out_docs, float* out_vals, uint32_t* out_len) {
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_loc_compact, dim3(n_jobs), dim3(64), 0, st, jobs, sorted_pairs, out_docs, out_vals, out_len);
}

}  // namespace vq

// ====================================================================================================
// k_union (K2) — multi-list union with per-doc max: materialises the hits of a leaf whose dictionary
// expansion matched many terms (resolve_token_to_anchor, search_field.rs:419-464: every posting becomes
// Hit(doc, term_score * (f16 / 100)), then sort by doc and dedup keeping the max).  One wave per span of the
// job's doc space; lane l walks list l (<= 64 lists per task, wider leaves are merged in two levels) through a
// 16-entry LDS window; each step takes the wave-wide minimum of the 64-bit heads (doc << 32 | ~order(score)),
// which yields the next doc AND its maximal score in one reduction.  Run twice: count, then write.
// ====================================================================================================
namespace vq {

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long x) {
#define VQ_MIN_STEP(ctrl, rm, bm)                                                                                       \
    {                                                                                                                   \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)x, ctrl, rm, bm, false);           \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(x >> 32), ctrl, rm, bm, false);  \
        const unsigned long long y = ((unsigned long long)hi << 32) | lo;                                               \
        x = y < x ? y : x;                                                                                              \
    }
    VQ_MIN_STEP(0x111, 0xF, 0xF)  // row_shr:1
    VQ_MIN_STEP(0x112, 0xF, 0xF)  // row_shr:2
    VQ_MIN_STEP(0x114, 0xF, 0xE)  // row_shr:4
    VQ_MIN_STEP(0x118, 0xF, 0xC)  // row_shr:8
    VQ_MIN_STEP(0x142, 0xA, 0xF)  // row_bcast:15
    VQ_MIN_STEP(0x143, 0xC, 0xF)  // row_bcast:31
#undef VQ_MIN_STEP
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t lane_lower_bound(const VQ_GLOBAL uint32_t* a, uint32_t n, uint32_t target) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Leaf hits around the entry anchors of a 1:n boost list with several values per anchor (compile.cpp emit_boost_1n, boost.rs:255-281): for the
// job's anchors a_0 < a_1 < ... the leaf's postings AT a_j (counts[2j]) and strictly between a_(j-1) and a_j (counts[2j + 1]; 0 for j = 0),
// summed over the leaf's lists.  A materialised leaf is one list: a lane per anchor, 64 anchors per block; otherwise a block per anchor with
// the lanes striding over the lists.  The host sends the anchors themselves (4 B each), not the ranges.
__global__ __launch_bounds__(64) void k_range_hits(const UList* __restrict__ ulists, const RangeJobD* __restrict__ jobs, uint32_t n_jobs,
                                                    const uint32_t* __restrict__ anchors, unsigned long long* __restrict__ counts) {
    uint32_t jl = 0, jh = n_jobs;
    while (jh - jl > 1u) {  // the job this block belongs to (uniform)
        const uint32_t mid = (jl + jh) >> 1;
        if (jobs[mid].block_begin <= blockIdx.x) jl = mid;
        else jh = mid;
    }
    const RangeJobD J = jobs[jl];
    const uint32_t b = blockIdx.x - J.block_begin;
    const VQ_GLOBAL uint32_t* an = as_global(anchors) + J.anchor_begin;
    if (J.n_lists == 1u) {
        const uint32_t j = b * 64u + threadIdx.x;
        if (j >= J.n_anchors) return;
        const UList L = ulists[J.list_begin];
        const VQ_GLOBAL uint32_t* d = as_global(L.docs);
        const uint32_t a = an[j];
        const uint32_t p1 = lane_lower_bound(d, L.len, a), p2 = lane_lower_bound(d, L.len, a + 1u);
        const uint32_t p0 = j ? lane_lower_bound(d, L.len, an[j - 1u] + 1u) : p1;
        counts[2ull * (J.anchor_begin + j)] = p2 - p1;
        counts[2ull * (J.anchor_begin + j) + 1ull] = p1 - p0;
        return;
    }
    if (b >= J.n_anchors) return;
    const uint32_t a = an[b], prev = b ? an[b - 1u] + 1u : a;
    unsigned long long hit = 0, between = 0;
    for (uint32_t i = threadIdx.x; i < J.n_lists; i += 64u) {
        const UList L = ulists[J.list_begin + i];
        const VQ_GLOBAL uint32_t* d = as_global(L.docs);
        const uint32_t p1 = lane_lower_bound(d, L.len, a);
        hit += lane_lower_bound(d, L.len, a + 1u) - p1;
        between += p1 - lane_lower_bound(d, L.len, prev);
    }
    for (uint32_t off = 32; off > 0; off >>= 1) {
        hit += shfl_u64(hit, (threadIdx.x + off) & 63u);
        between += shfl_u64(between, (threadIdx.x + off) & 63u);
    }
    if (threadIdx.x == 0) {
        counts[2ull * (J.anchor_begin + b)] = hit;
        counts[2ull * (J.anchor_begin + b) + 1ull] = between;
    }
}
void launch_range_hits(hipStream_t st, uint32_t n_blocks, uint32_t n_jobs, const UList* ulists, const RangeJobD* jobs, const uint32_t* anchors, unsigned long long* counts) {
    if (!n_blocks || !n_jobs) return;
    hipLaunchKernelGGL(k_range_hits, dim3(n_blocks), dim3(64), 0, st, ulists, jobs, n_jobs, anchors, counts);
}

constexpr uint32_t kUnionWindow = 16;

template <bool WRITE>
__global__ __launch_bounds__(64) void k_union(const UList* __restrict__ ulists, const UTask* __restrict__ tasks, const uint32_t* __restrict__ span_task,
                                              uint32_t* __restrict__ span_cnt, const uint64_t* __restrict__ span_off, uint32_t* __restrict__ out_docs,
                                              float* __restrict__ out_vals, uint32_t* __restrict__ task_min) {
    __shared__ unsigned long long win[kUnionWindow][64];
    const uint32_t span = blockIdx.x, lane = threadIdx.x;
    const UTask task = tasks[span_task[span]];
    const uint32_t s = span - task.span_begin;
    // span bounds in doc space: quantiles of the task's longest list
    uint32_t lo_doc = 0u, hi_doc = 0xFFFFFFFFu;
    {
        const UList piv = ulists[task.pivot];
        const VQ_GLOBAL uint32_t* pd = as_global(piv.docs);
        if (s > 0) lo_doc = pd[(unsigned long long)piv.len * s / task.n_spans];
        if (s + 1 < task.n_spans) hi_doc = pd[(unsigned long long)piv.len * (s + 1) / task.n_spans];
    }
    UList L{};
    uint32_t pos = 0, end = 0;
    if (lane < task.n_lists) {
        L = ulists[task.list_begin + lane];
        const VQ_GLOBAL uint32_t* d = as_global(L.docs);
        pos = lane_lower_bound(d, L.len, lo_doc);
        end = hi_doc == 0xFFFFFFFFu ? L.len : lane_lower_bound(d, L.len, hi_doc);
    }
    const VQ_GLOBAL uint32_t* docs = as_global(L.docs);
    const VQ_GLOBAL uint16_t* s16 = as_global(reinterpret_cast<const uint16_t*>(L.scores));
    const VQ_GLOBAL float* s32 = as_global(reinterpret_cast<const float*>(L.scores));
    const bool f32 = L.flags & 1u;
    uint32_t wbase = pos;  // list index of window slot 0
    auto refill = [&]() {
        wbase = pos;
#pragma unroll
        for (uint32_t k = 0; k < kUnionWindow; ++k) {
            const uint32_t i = pos + k;
            if (i < end) {
                const float v = f32 ? s32[i] : posting_value(L.term_score, s16[i]);
                win[k][lane] = ((unsigned long long)docs[i] << 32) | (uint32_t)~order_f32(__float_as_uint(v));
            }
        }
    };
    unsigned long long head = ~0ull;
    if (pos < end) {
        refill();
        head = win[0][lane];
    }
    const uint64_t base = WRITE ? span_off[span] : 0ull;
    uint32_t n = 0;
    uint32_t best = 0xFFFFFFFFu;  // min of ~order(value) over the emitted docs == the largest value of the merged list
    unsigned long long pending = 0ull;
    while (true) {
        const unsigned long long m = wave_min_u64(head);
        if (m == ~0ull) break;
        if (WRITE) best = (uint32_t)m < best ? (uint32_t)m : best;
        if (WRITE) {
            if ((n & 63u) == lane) pending = m;
            if ((n & 63u) == 63u) {
                out_docs[base + (n - 63u) + lane] = (uint32_t)(pending >> 32);
                out_vals[base + (n - 63u) + lane] = __uint_as_float(unorder_f32(~(uint32_t)pending));
            }
        }
        ++n;
        if ((head >> 32) == (m >> 32)) {  // every list holding this doc moves on
            ++pos;
            if (pos >= end) head = ~0ull;
            else {
                if (pos - wbase >= kUnionWindow) refill();
                head = win[pos - wbase][lane];
            }
        }
    }
    if (WRITE) {
        const uint32_t done = n & ~63u;
        if (lane < (n & 63u)) {
            out_docs[base + done + lane] = (uint32_t)(pending >> 32);
            out_vals[base + done + lane] = __uint_as_float(unorder_f32(~(uint32_t)pending));
        }
        if (lane == 0 && best != 0xFFFFFFFFu) atomicMin(&task_min[span_task[span]], best);
        if (s + 1 == task.n_spans && lane < 8u) {  // list padding: sentinel docs up to the vector width and beyond
            out_docs[base + n + lane] = 0xFFFFFFFFu;
            out_vals[base + n + lane] = 0.0f;
        }
    } else if (lane == 0) {
        span_cnt[span] = n;
    }
}

void launch_union(hipStream_t st, bool write, uint32_t total_spans, const UList* ulists, const UTask* tasks, const uint32_t* span_task, uint32_t* span_cnt,
                  const uint64_t* span_off, uint32_t* out_docs, float* out_vals, uint32_t* task_min) {
    if (!total_spans) return;
    if (write) hipLaunchKernelGGL(k_union<true>, dim3(total_spans), dim3(64), 0, st, ulists, tasks, span_task, span_cnt, span_off, out_docs, out_vals, task_min);
    else hipLaunchKernelGGL(k_union<false>, dim3(total_spans), dim3(64), 0, st, ulists, tasks, span_task, span_cnt, span_off, out_docs, out_vals, task_min);
}

}  // namespace vq

// ====================================================================================================
// k_scan_union — pure simple queries whose hits are dense: a single leaf (K1 streaming scan) or one OR over
// 2..4 single-list posting leaves (K4).  Every posting is a hit, so the f16 scores are streamed with the doc
// ids (coalesced 16 B + 8 B per lane) instead of being gathered per survivor:
//   n == 1: the span's slice of the list is streamed once; score, key, threshold test per posting.
//   OR:     tiles of 2048 docs.  Pass 1 scatters every list's in-tile postings into LDS (presence bit + raw f16 at the
//           doc's offset).  Pass 2 streams the same postings again (L2 hits); a posting is evaluated by the FIRST list
//           that holds its doc: lanes work on postings, not on bitmap bits, so there are no per-lane bit loops.
// Same arithmetic as simple_flush (set_op.rs:169-186), bit-identical results.
// ====================================================================================================
namespace vq {

#ifndef VQ_UT
#define VQ_UT 2048
#endif
constexpr uint32_t kUT = VQ_UT;       // docs per tile
constexpr uint32_t kUTW = kUT / 32;   // bitmap words per list and tile
// LDS map (u32): misc[8] | cand[2*cand_cap] | bm[4][kUTW] | val u16 [4][kUT]
size_t scan_union_lds_bytes(uint32_t cand_cap, bool with_or) { return (size_t)(8 + 2 * cand_cap + (with_or ? 4 * kUTW + 4 * kUT / 2 : 0)) * 4 + 16; }

__device__ __forceinline__ void union_push(bool pending, unsigned long long key, const CandState& cs, uint32_t top_k) {
    while (true) {  // uniform
        if (pending) {
            if (key > *cs.thr) {
                const uint32_t pos = atomicAdd(cs.n, 1u);
                if (pos < cs.cap) {
                    cs.cand[pos] = key;
                    pending = false;
                }
            } else pending = false;
        }
        if (!__syncthreads_or(pending ? 1 : 0)) break;
        cand_prune(cs, top_k);
    }
}

__global__ __launch_bounds__(64) void k_scan_union(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                   const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq, uint32_t cand_cap,
                                                   unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t ql;
    {
        uint32_t lo = 0, hi = nq;
        const uint32_t wg = blockIdx.x;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (span_base[mid] <= wg) lo = mid;
            else hi = mid;
        }
        ql = lo;
    }
    const uint32_t span = blockIdx.x - span_base[ql];
    const uint32_t q = qmap[ql];
    const uint8_t* blob = blobs + blob_off[q];
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const uint32_t n = H->simple_n;
    const uint32_t top_k = H->top_k;
    const DList* gl = reinterpret_cast<const DList*>(blob + H->off_lists);
    const DOp* gops = reinterpret_cast<const DOp*>(blob + H->off_ops);

    const VQ_GLOBAL uint32_t* docs[4] = {nullptr, nullptr, nullptr, nullptr};
    const uint32_t* docs_flat[4] = {nullptr, nullptr, nullptr, nullptr};
    const VQ_GLOBAL uint16_t* scs[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t len[4] = {0, 0, 0, 0};
    float ts[4] = {0.f, 0.f, 0.f, 0.f};
    uint8_t slot[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) {
            const DList& d = gl[gops[k].list_begin];
            docs[k] = as_global(d.docs);
            docs_flat[k] = d.docs;
            scs[k] = as_global(d.scores);
            len[k] = d.len;
            ts[k] = d.term_score;
        }
    uint32_t nslots = 1;
    if (n > 1) {
        nslots = gops[n].nslots;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) slot[k] = gops[n].child_slot[k];
    }

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* cand_n = lds + 2;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + 8);
    uint32_t* bm = lds + 8 + 2 * cand_cap;                                  // [4][kUTW]
    uint16_t* val = reinterpret_cast<uint16_t*>(bm + 4 * kUTW);            // [4][kUT]
    CandState cs{cand, cand_n, thr, cand_cap};
    cs.upper = H->key_upper;

    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const bool cov_mode = n == 1u && ((H->simple_flags >> 28) & 1u) != 0u;  // the single leaf streams its tile-packed image (spans end on its tiles)
    const uint32_t span_mask = cov_mode ? ~((1u << kProbeTileShift) - 1u) : ~(kSW - 1u);
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & span_mask);
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & span_mask);
    const uint32_t keys_base = H->keys_base;

    uint32_t cur[4] = {0, 0, 0, 0}, end[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (k < n) {
            cur[k] = wave_lower_bound(docs_flat[k], len[k], span_lo);
            end[k] = cur[k] + wave_lower_bound(docs_flat[k] + cur[k], len[k] - cur[k], span_hi);
        }
    if (lane == 0) {
        *thr = 0ull;
        *cand_n = 0;
    }
    __syncthreads();
    unsigned long long hits = 0;
    const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

    if (n == 1) {
        auto stream_postings = [&](const uint32_t c0, const uint32_t e0) {  // uniform bounds: postings [c0, e0) of the list, ids and scores
        if (e0 <= c0) return;
        // ---- K1: stream [cur, end) of the one list, kU1 vectors of 4 postings per lane and round, the next round in flight.
        // With a positive term score the value is monotone in the raw f16 bits, so postings are pre-filtered by comparing the raw
        // bits with `raw_min`, the smallest raw score that can still reach the threshold; only those build a key.
        const VQ_GLOBAL u32x4* dptr = reinterpret_cast<const VQ_GLOBAL u32x4*>(docs[0]);
        const VQ_GLOBAL uint2* sptr = reinterpret_cast<const VQ_GLOBAL uint2*>(scs[0]);
        const uint32_t v_end = (e0 + 3u) >> 2;
        constexpr uint32_t kU1 = 3;
        u32x4 d4[kU1];
        uint2 s4[kU1];
        const uint32_t v_first = c0 >> 2;
        // rounds run from the END of the slice to its start: under (score desc, id desc) a later posting wins every tie, so an
        // ascending scan of a list with few distinct scores would push on every tie; descending, ties never beat the threshold
        const uint32_t n_rounds = (v_end - v_first + kU1 * 64u - 1u) / (kU1 * 64u);
#pragma unroll
        for (uint32_t u = 0; u < kU1; ++u) {
            const uint32_t v = v_first + (n_rounds ? (n_rounds - 1u) * kU1 * 64u : 0u) + u * 64u + lane;
            d4[u] = kSent;
            s4[u] = uint2{0u, 0u};
            if (v < v_end) {
                d4[u] = dptr[v];
                s4[u] = uint2{sptr[v].x, sptr[v].y};
            }
        }
        const bool monotone = ts[0] > 0.0f;  // otherwise every posting takes the exact path (raw_min = 0)
        unsigned long long thr_seen = ~0ull;
        uint32_t raw_min = 0;
        for (uint32_t r = n_rounds; r-- > 0;) {  // uniform
            const uint32_t v0 = v_first + r * kU1 * 64u;
            u32x4 nd4[kU1];
            uint2 ns4[kU1];
#pragma unroll
            for (uint32_t u = 0; u < kU1; ++u) {
                const uint32_t vn = v0 - kU1 * 64u + u * 64u + lane;  // the round before (only read when r > 0)
                nd4[u] = kSent;
                ns4[u] = uint2{0u, 0u};
                if (r > 0 && vn < v_end) {
                    nd4[u] = dptr[vn];
                    ns4[u] = uint2{sptr[vn].x, sptr[vn].y};
                }
            }
            const unsigned long long thr_reg = *thr;
            if (thr_reg != thr_seen) {  // uniform: the threshold rose — smallest positive-f16 bit pattern whose value reaches its score
                thr_seen = thr_reg;
                raw_min = 0;
                if (monotone && thr_reg != 0ull) {
                    const uint32_t tbits = (uint32_t)(thr_reg >> 32);
                    uint32_t lo = 0, hi = 0x7C00u;  // [0, +inf): finite non-negative f16
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (order_f32(__float_as_uint(posting_value_fast(ts[0], (uint16_t)mid))) < tbits) lo = mid + 1;
                        else hi = mid;
                    }
                    raw_min = lo;
                }
            }
            // interior rounds need no index checks
            const bool interior = (v0 << 2) >= c0 && ((v0 + kU1 * 64u) << 2) <= e0;
            bool any = false;
            bool pass[kU1 * 4];
#pragma unroll
            for (uint32_t u = 0; u < kU1; ++u) {
                const uint32_t v = v0 + u * 64u + lane;
                const uint32_t i0 = v << 2;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t raw = j == 0 ? (s4[u].x & 0xFFFFu) : j == 1 ? (s4[u].x >> 16) : j == 2 ? (s4[u].y & 0xFFFFu) : (s4[u].y >> 16);
                    // raw >= raw_min as integers: true for every candidate (negative / NaN patterns compare high and take the exact path)
                    bool p = raw >= raw_min;
                    if (!interior) p = p && v < v_end && (i0 + j) >= c0 && (i0 + j) < e0;
                    pass[u * 4 + j] = p;
                    any = any || p;
                }
            }
            if (__ballot(any)) {  // uniform; rare once the threshold has risen
#pragma unroll
                for (uint32_t u = 0; u < kU1; ++u) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        if (__ballot(pass[u * 4 + j])) {  // uniform
                            const uint32_t raw = j == 0 ? (s4[u].x & 0xFFFFu) : j == 1 ? (s4[u].x >> 16) : j == 2 ? (s4[u].y & 0xFFFFu) : (s4[u].y >> 16);
                            const float score = posting_value_fast(ts[0], (uint16_t)raw);
                            const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)comp4(d4[u], j);
                            union_push(pass[u * 4 + j] && key > *thr && key < cs.upper, key, cs, top_k);
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kU1; ++u) {
                d4[u] = nd4[u];
                s4[u] = ns4[u];
            }
        }
        };
        hits = end[0] - cur[0];
        if (!cov_mode) stream_postings(cur[0], end[0]);
        else {
            // The list's tile-packed image (cov32: in-tile offset << 16 | f16 score, 4 B per posting instead of 6) is streamed instead of its ids and
            // scores.  A packed word says which doc it is only together with its tile, so the few postings that pass the raw-score test look their
            // tile up in the directory (a binary search per lane, rare once the threshold stands); the span's LAST tiles — about 2048 postings — still
            // run on the id stream and set that threshold first.  Without one (fewer hits than the request ranks) everything stays on the id stream.
            const uint32_t* tdir = gl[gops[0].list_begin].tile_dir;
            const uint32_t* gdir = as_const<DProbe>(blob + H->off_simple2)->leaf[0].gdir;
            const VQ_GLOBAL u32x4* cc4 = as_global(reinterpret_cast<const u32x4*>(as_const<DProbe>(blob + H->off_simple2)->leaf[0].cov32));
            const uint32_t bitmap_base = H->bitmap_base;
            const uint32_t t_first = (span_lo - bitmap_base) >> kProbeTileShift;
            const uint32_t t_end = span_hi > span_lo ? ((span_hi - 1u - bitmap_base) >> kProbeTileShift) + 1u : t_first;
            uint32_t t_switch = t_end;
            {
                const uint32_t warm = top_k * 4u > 2048u ? top_k * 4u : 2048u;  // (the candidate buffer holds 2 x top_k keys before its first prune sets a threshold)
                const uint32_t want = end[0] - cur[0] > warm ? end[0] - warm : cur[0];
                // the 32768-doc tile that holds posting `want` (directory: one entry per 16384 docs, entries below a boundary): from its start on
                // — at least 2048 postings, at most a tile more — the id stream runs
                const uint32_t i = wave_lower_bound(tdir + 2u * t_first, 2u * (t_end - t_first) + 1u, want + 1u);  // first boundary with more than `want` postings below it
                t_switch = t_first + ((i ? i - 1u : 0u) >> 1);
                t_switch = t_switch < t_end ? t_switch : t_end;
            }
            const uint32_t split = as_global(tdir)[2u * t_switch];  // postings below tile t_switch
            stream_postings(split > cur[0] ? split : cur[0], end[0]);
            if (t_switch > t_first) {  // uniform
                __syncthreads();
                if (*thr == 0ull) stream_postings(cur[0], split < end[0] ? split : end[0]);  // uniform: no threshold yet
                else {
                    const uint32_t g_first = as_global(gdir)[t_first], g_end = as_global(gdir)[t_switch];
                    const uint32_t v_first = g_first * 2u, v_end = g_end * 2u;  // 16-byte vectors of 4 packed words
#ifndef VQ_UC
#define VQ_UC 4
#endif
                    constexpr uint32_t kUc = VQ_UC;
                    const uint32_t n_rounds = (v_end - v_first + kUc * 64u - 1u) / (kUc * 64u);
                    const u32x4 kPad = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                    u32x4 e4[kUc];
#pragma unroll
                    for (uint32_t u = 0; u < kUc; ++u) {
                        const uint32_t v = v_first + (n_rounds ? (n_rounds - 1u) * kUc * 64u : 0u) + u * 64u + lane;
                        e4[u] = v < v_end ? cc4[v] : kPad;
                    }
                    unsigned long long thr_seen = ~0ull;
                    uint32_t raw_min = 0;
                    for (uint32_t r = n_rounds; r-- > 0;) {  // uniform; descending like the id stream (ties never beat the threshold)
                        const uint32_t v0 = v_first + r * kUc * 64u;
                        u32x4 ne4[kUc];
#pragma unroll
                        for (uint32_t u = 0; u < kUc; ++u) {
                            const uint32_t vn = v0 - kUc * 64u + u * 64u + lane;  // the round before (only read when r > 0)
                            ne4[u] = (r > 0 && vn >= v_first && vn < v_end) ? cc4[vn] : kPad;
                        }
                        const unsigned long long thr_reg = *thr;
                        if (thr_reg != thr_seen) {  // uniform: the threshold rose — smallest positive-f16 bit pattern whose value EXCEEDS its score.
                            // (A posting that merely ties the threshold's score loses on the doc id without its doc being known: the threshold as
                            //  read at the top of a round holds keys of earlier rounds only — docs above every doc of this round.  With the few
                            //  distinct f16 scores of a list, ties are most of what reaches the threshold's score.)
                            thr_seen = thr_reg;
                            const uint32_t tbits = (uint32_t)(thr_reg >> 32);
                            uint32_t lo = 0, hi = 0x7C00u;
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (order_f32(__float_as_uint(posting_value_fast(ts[0], (uint16_t)mid))) <= tbits) lo = mid + 1;
                                else hi = mid;
                            }
                            raw_min = lo;
                        }
                        bool any = false;
#pragma unroll
                        for (uint32_t u = 0; u < kUc; ++u) {
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                const uint32_t e = comp4(e4[u], j);
                                any = any || ((e & 0xFFFFu) >= raw_min && e != 0xFFFFFFFFu);
                            }
                        }
                        if (__ballot(any)) {  // uniform; rare
#pragma unroll
                            for (uint32_t u = 0; u < kUc; ++u) {
#pragma unroll
                                for (uint32_t j = 0; j < 4; ++j) {
                                    const uint32_t e = comp4(e4[u], j);
                                    const bool p = (e & 0xFFFFu) >= raw_min && e != 0xFFFFFFFFu;
                                    if (__ballot(p)) {  // uniform
                                        unsigned long long key = 0ull;
                                        if (p) {
                                            // the tile of granule g: the last directory entry <= g among tiles [t_first, t_switch)
                                            const uint32_t g = (v0 + u * 64u + lane) >> 1;
                                            uint32_t lo = t_first, hi = t_switch;  // invariant: gdir[lo] <= g < gdir[hi]
                                            while (hi - lo > 1u) {
                                                const uint32_t mid = (lo + hi) >> 1;
                                                if (as_global(gdir)[mid] <= g) lo = mid;
                                                else hi = mid;
                                            }
                                            const uint32_t doc = bitmap_base + (lo << kProbeTileShift) + (e >> 16);
                                            const float score = posting_value_fast(ts[0], (uint16_t)(e & 0xFFFFu));
                                            key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
                                        }
                                        union_push(p && key > *thr && key < cs.upper, key, cs, top_k);
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (uint32_t u = 0; u < kUc; ++u) e4[u] = ne4[u];
                    }
                }
            }
        }
    } else {
        // ---- K4: OR over n lists, tile by tile
        bool ident = nslots == n;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n && slot[k] != k) ident = false;
        uint32_t nxt[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < n && cur[k] < end[k]) nxt[k] = docs[k][cur[k]];
        while (true) {
            uint32_t head = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < n) head = nxt[k] < head ? nxt[k] : head;
            if (head >= span_hi) break;
            const uint32_t tile_lo = head & ~(kUT - 1u);
            const uint32_t tile_end = tile_lo + kUT;
            const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;
            uint32_t c1[4] = {0, 0, 0, 0};
            // pass 1: presence bits + raw scores of every list's in-tile postings
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < n) bm[k * kUTW + lane] = 0u;
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < n) {
                    c1[k] = cur[k];
                    if (nxt[k] < tile_hi) {  // uniform
                        const VQ_GLOBAL u32x4* dptr = reinterpret_cast<const VQ_GLOBAL u32x4*>(docs[k]);
                        const VQ_GLOBAL uint2* sptr = reinterpret_cast<const VQ_GLOBAL uint2*>(scs[k]);
                        const uint32_t v_end = (end[k] + 3u) >> 2;
                        uint32_t* bmk = bm + k * kUTW;
                        uint16_t* valk = val + k * kUT;
                        uint32_t total = 0, boundary = 0xFFFFFFFFu;
                        for (uint32_t v0 = cur[k] >> 2;; v0 += 64u) {  // uniform
                            const uint32_t v = v0 + lane;
                            u32x4 d4 = kSent;
                            uint2 s4 = uint2{0u, 0u};
                            if (v < v_end) {
                                d4 = dptr[v];
                                s4 = uint2{sptr[v].x, sptr[v].y};
                            }
                            const uint32_t i0 = v << 2;
                            uint32_t mine = 0, first_out = 0xFFFFFFFFu;
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                const uint32_t d = comp4(d4, j);
                                const uint32_t i = i0 + j;
                                const bool live = v < v_end && i >= cur[k] && i < end[k];
                                const bool in = live && d < tile_hi;
                                if (in) {
                                    const uint32_t o = d - tile_lo;
                                    const uint32_t raw = j == 0 ? (s4.x & 0xFFFFu) : j == 1 ? (s4.x >> 16) : j == 2 ? (s4.y & 0xFFFFu) : (s4.y >> 16);
                                    atomicOr(&bmk[o >> 5], 1u << (o & 31u));
                                    valk[o] = (uint16_t)raw;
                                    ++mine;
                                } else if (live && first_out == 0xFFFFFFFFu) first_out = d;
                            }
                            uint32_t round_in;
                            (void)wave_excl_scan_u32(mine, &round_in);
                            total += round_in;
                            // the first live posting outside the tile (sorted list: the minimum over the lanes)
                            const unsigned long long outm = __ballot(first_out != 0xFFFFFFFFu);
                            if (outm) {
                                boundary = (uint32_t)__builtin_amdgcn_readlane((int)first_out, (int)(__ffsll((long long)outm) - 1));
                                break;
                            }
                            if (v0 + 64u >= v_end) break;  // list exhausted inside the tile
                        }
                        c1[k] = cur[k] + total;
                        nxt[k] = boundary;
                    }
                }
            }
            __syncthreads();
            // pass 2: evaluate every posting whose doc no earlier list holds
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < n && c1[k] > cur[k]) {  // uniform
                    const VQ_GLOBAL u32x4* dptr = reinterpret_cast<const VQ_GLOBAL u32x4*>(docs[k]);
                    const VQ_GLOBAL uint2* sptr = reinterpret_cast<const VQ_GLOBAL uint2*>(scs[k]);
                    const uint32_t v_end = (c1[k] + 3u) >> 2;
                    for (uint32_t v0 = cur[k] >> 2; v0 < v_end; v0 += 64u) {  // uniform
                        const uint32_t v = v0 + lane;
                        u32x4 d4 = kSent;
                        uint2 s4 = uint2{0u, 0u};
                        if (v < v_end) {
                            d4 = dptr[v];
                            s4 = uint2{sptr[v].x, sptr[v].y};
                        }
                        const uint32_t i0 = v << 2;
                        const unsigned long long thr_reg = *thr;
                        unsigned long long key[4];
                        bool pend[4];
                        uint32_t fresh = 0;
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            const uint32_t d = comp4(d4, j);
                            const uint32_t i = i0 + j;
                            bool live = v < v_end && i >= cur[k] && i < c1[k];
                            const uint32_t o = live ? d - tile_lo : 0u;
                            const uint32_t raw = j == 0 ? (s4.x & 0xFFFFu) : j == 1 ? (s4.x >> 16) : j == 2 ? (s4.y & 0xFFFFu) : (s4.y >> 16);
                            float vals[4] = {0.f, 0.f, 0.f, 0.f};
                            uint32_t pm = 1u << k;
#pragma unroll
                            for (uint32_t i2 = 0; i2 < 4; ++i2) {
                                if (i2 < n && i2 != k) {
                                    const bool bit = (bm[i2 * kUTW + (o >> 5)] >> (o & 31u)) & 1u;
                                    if (bit) {
                                        if (i2 < k) live = false;  // an earlier list evaluates this doc
                                        else {
                                            pm |= 1u << i2;
                                            vals[i2] = posting_value_fast(ts[i2], val[i2 * kUT + o]);
                                        }
                                    }
                                }
                            }
                            const float own = posting_value_fast(ts[k], (uint16_t)raw);
                            if (k == 0) vals[0] = own;
                            else if (k == 1) vals[1] = own;
                            else if (k == 2) vals[2] = own;
                            else vals[3] = own;
                            float sum = 0.0f, nd = 0.0f;  // set_op.rs:169-186
                            if (ident) {  // operand k is slot k: absent operands contribute max(0, -) = 0
#pragma unroll
                                for (uint32_t i2 = 0; i2 < 4; ++i2)
                                    if (i2 < n) {
                                        const float m = fmaxf(0.0f, vals[i2]);  // vals[i2] == 0 when absent
                                        nd += m >= 0.00001f ? 1.0f : 0.0f;
                                        sum += m;
                                    }
                            } else {
                                for (uint32_t sl = 0; sl < nslots; ++sl) {
                                    float m = 0.0f;
#pragma unroll
                                    for (uint32_t i2 = 0; i2 < 4; ++i2)
                                        if (i2 < n && slot[i2] == sl && ((pm >> i2) & 1u)) m = fmaxf(m, vals[i2]);
                                    if (m >= 0.00001f) nd += 1.0f;
                                    sum += m;
                                }
                            }
                            const float score = sum * nd * nd;
                            key[j] = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)d;
                            fresh += live ? 1u : 0u;
                            pend[j] = live && key[j] > thr_reg && key[j] < cs.upper;
                        }
                        uint32_t round_fresh;
                        (void)wave_excl_scan_u32(fresh, &round_fresh);
                        hits += round_fresh;
                        if (__ballot(pend[0] || pend[1] || pend[2] || pend[3])) {  // uniform
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) union_push(pend[j], key[j], cs, top_k);
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) cur[k] = c1[k];
            __syncthreads();
        }
    }
    __syncthreads();
    cand_prune(cs, top_k);
    {
        const uint32_t cn = *cand_n;
        unsigned long long* out = span_keys + (size_t)keys_base + (size_t)span * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cand[i] : 0ull;
    }
    if (lane == 0 && hits) atomicAdd(&num_hits[q], hits);
}

void launch_scan_union(hipStream_t st, bool with_or, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                       const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits) {
    if (!total_spans) return;
    // single-leaf queries need no tile arrays: with 2-3 KB of LDS the register file, not LDS, bounds the waves per CU
    hipLaunchKernelGGL(k_scan_union, dim3(total_spans), dim3(64), scan_union_lds_bytes(cand_cap, with_or), st, blobs, blob_off, span_base, qmap, nq, cand_cap, span_keys,
                       num_hits);
}


// ====================================================================================================
// k_scan_leaf_f32 — a query that is ONE materialised leaf (k_union output: doc ids + final f32 values) with no filter and no
// score-shaping sink stage: every entry is a hit, its value is its score.  One wave per span of the LIST (entries, not doc ids:
// perfectly even work), entries streamed from the end of the slice (under "score desc, id desc" later entries win ties, so a
// descending walk never pushes on a tie); facets, if any, are counted for every entry.  This is the fuzzy / prefix single-term
// request ("search as you type"), which the tile kernels served at a few G entries/s through their bitmap machinery.
// ====================================================================================================
__global__ __launch_bounds__(64) void k_scan_leaf_f32(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                      const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq,
                                                      uint32_t cand_cap, unsigned long long* __restrict__ span_keys,
                                                      unsigned long long* __restrict__ num_hits, uint32_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t ql;
    {
        uint32_t lo = 0, hi = nq;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (span_base[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        ql = lo;
    }
    const uint32_t q = qmap[ql];
    const uint32_t span = blockIdx.x - span_base[ql];
    const uint8_t* blob = blobs + blob_off[q];
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const DList& L = reinterpret_cast<const DList*>(blob + H->off_lists)[reinterpret_cast<const DOp*>(blob + H->off_ops)[0].list_begin];
    const DFacet* facets = reinterpret_cast<const DFacet*>(blob + H->off_facets);
    const uint32_t n_facets = H->n_facets, top_k = H->top_k, n_spans = H->n_spans;
    const VQ_GLOBAL uint32_t* docs = as_global(L.docs);
    const VQ_GLOBAL uint32_t* vals = as_global(reinterpret_cast<const uint32_t*>(L.scores));

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* cand_n = lds + 2;
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + 8);
    CandState cs{cand, cand_n, thr, cand_cap, reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blob) + offsetof(QHeader, gthr))};
    cs.upper = H->key_upper;
    if (lane == 0) {
        *thr = 0ull;
        *cand_n = 0;
    }
    __syncthreads();
    // this span's slice of the list: entries [lo, hi)
    const uint32_t lo = (uint32_t)((unsigned long long)L.len * span / n_spans), hi = (uint32_t)((unsigned long long)L.len * (span + 1) / n_spans);
    uint32_t round = 0;
    uint32_t my_gb = 0;
    for (uint32_t top = hi; top > lo; top = top - lo > 64u ? top - 64u : lo) {  // uniform; descending
        const uint32_t base = top - lo > 64u ? top - 64u : lo;
        const uint32_t i = base + lane;
        const bool have = i < top;
        uint32_t doc = 0, bits = 0;
        if (have) {
            doc = docs[i];
            bits = vals[i];
        }
        if ((round++ & 7u) == 0u && lane == 0) {  // adopt the threshold other spans of the query have published
            const unsigned long long g = *reinterpret_cast<volatile unsigned long long*>(cs.gthr);
            if (g > *thr) *thr = g;
        }
        for (uint32_t f = 0; f < n_facets; ++f) {  // persistence.rs:164-175 count_values_for_ids: every hit counts
            const DFacet& fa = facets[f];
            if (have && doc >= fa.key_base && doc - fa.key_base < fa.num_keys) {
                const uint32_t row = doc - fa.key_base;
                if (fa.direct) {
                    const uint32_t v = as_global(fa.direct)[row];
                    my_gb += 4u;
                    if (v < fa.num_values) atomicAdd(&hist[fa.hist_off + v], 1u);
                } else {
                    const unsigned long long e0 = as_global(fa.offsets)[row], e1 = as_global(fa.offsets)[row + 1];
                    my_gb += 16u + 4u * (uint32_t)(e1 - e0);
                    for (unsigned long long e = e0; e < e1; ++e) {
                        const uint32_t v = as_global(fa.values)[e];
                        if (v < fa.num_values) atomicAdd(&hist[fa.hist_off + v], 1u);
                    }
                }
            }
        }
        const unsigned long long key = ((unsigned long long)order_f32(bits) << 32) | (unsigned long long)doc;
        const bool pend = have && key > *thr && key < cs.upper;
        if (__ballot(pend)) union_push(pend, key, cs, top_k);  // uniform
    }
    cand_prune(cs, top_k);
    {
        const uint32_t cn = *cand_n;
        unsigned long long* out = span_keys + (size_t)H->keys_base + (size_t)span * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cand[i] : 0ull;
    }
    if (lane == 0 && hi > lo) atomicAdd(&num_hits[q], (unsigned long long)(hi - lo));
    if (n_facets) {  // uniform
        uint32_t gb_total;
        (void)wave_excl_scan_u32(my_gb, &gb_total);
        if (lane == 0 && gb_total && H->stat_off) atomicAdd(&num_hits[H->stat_off], (unsigned long long)gb_total);
    }
}
void launch_scan_leaf_f32(hipStream_t st, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base, const uint32_t* qmap,
                          uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist) {
    if (!total_spans) return;
    hipLaunchKernelGGL(k_scan_leaf_f32, dim3(total_spans), dim3(64), (8 + 2 * (size_t)cand_cap) * 4 + 16, st, blobs, blob_off, span_base, qmap, nq, cand_cap,
                       span_keys, num_hits, hist);
}

// ------------------------------------------------------------------------------------ explain
// One lane per returned hit: the request's score tree evaluated for that one doc from the posting lists of every matched term (binary search:
// tens of docs, no tiles), every value the reference's Explain records quote written to the doc's trace — TermToAnchor's anchor and final score
// (search_field.rs:426-437), OrSumOverDistinctTerms (set_op.rs:190), Boost (boost.rs:297-300, 371-374).  The host only formats the records.
__global__ __launch_bounds__(64) void k_explain(uint32_t n_docs, const ExQuery* __restrict__ queries, const uint32_t* __restrict__ doc_query,
                                                const uint32_t* __restrict__ docs, const ExOp* __restrict__ ops, const uint16_t* __restrict__ aux,
                                                const ExList* __restrict__ lists, const DColBoost* __restrict__ cols, uint32_t* __restrict__ trace) {
    const uint32_t d = blockIdx.x * 64u + threadIdx.x;
    if (d >= n_docs) return;
    const ExQuery Q = queries[doc_query[d]];
    const uint32_t doc = docs[d];
    uint32_t* T = trace + Q.trace_begin + (size_t)(d - Q.doc_begin) * explain_trace_words(Q.n_lists, Q.n_ops, Q.n_col);
    uint32_t* T_ops = T + 3u * Q.n_lists;
    uint32_t* T_cols = T_ops + 3u * Q.n_ops;
    float stack[kExStack];
    bool pres[kExStack];
    uint32_t sp = 0;
    for (uint32_t o = 0; o < Q.n_ops; ++o) {
        const ExOp op = ops[Q.op_begin + o];
        float s = 0.0f, extra = 0.0f;
        bool present = false;
        if (op.kind == XP_LEAF) {
            for (uint32_t j = 0; j < op.b; ++j) {
                const ExList L = lists[Q.list_begin + op.a + j];
                uint32_t lo = 0, hi = L.len;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (L.docs[mid] < doc) lo = mid + 1u;
                    else hi = mid;
                }
                uint32_t* t = T + 3u * (op.a + j);
                if (lo < L.len && L.docs[lo] == doc) {
                    const uint16_t raw = L.scores[lo];
                    const float anchor = __half2float(__ushort_as_half(raw)) / 100.0f;  // search_field.rs:426
                    const float v = L.term_score * anchor;
                    t[0] = raw;
                    t[1] = __float_as_uint(anchor);
                    t[2] = __float_as_uint(v);
                    if (!present || v > s) s = v;  // dedup keeps the max (search_field.rs:455-461)
                    present = true;
                } else {
                    t[0] = 0xFFFFFFFFu;
                    t[1] = 0u;
                    t[2] = 0u;
                }
            }
        } else if (op.kind == XP_AND) {
            const uint32_t base = sp - op.nchild;
            present = true;
            for (uint32_t k = 0; k < op.nchild; ++k) present = present && pres[base + k];
            if (present)
                for (uint32_t k = 0; k < op.nchild; ++k) s += stack[base + aux[op.a + k]];  // set_op.rs:415-416, the shortest operand last
            sp = base;
        } else {
            const uint32_t base = sp - op.nchild;
            float sum = 0.0f, nd = 0.0f;
            for (uint32_t slot = 0; slot < op.b; ++slot) {  // set_op.rs:169-186
                float m = 0.0f;
                for (uint32_t k = 0; k < op.nchild; ++k)
                    if (aux[op.a + k] == slot && pres[base + k]) {
                        present = true;
                        m = fmaxf(m, stack[base + k]);
                    }
                if (m >= 0.00001f) nd += 1.0f;
                sum += m;
            }
            s = sum * nd * nd;
            extra = sum;
            sp = base;
        }
        T_ops[3u * o] = present ? 1u : 0u;
        T_ops[3u * o + 1u] = __float_as_uint(s);
        T_ops[3u * o + 2u] = __float_as_uint(extra);
        stack[sp] = s;
        pres[sp] = present;
        ++sp;
    }
    const bool root = sp != 0 && pres[0];
    float score = sp != 0 ? stack[0] : 0.0f;
    const float tree_score = score;
    for (uint32_t k = 0; k < Q.n_col; ++k) {  // add_boost (boost.rs:470-504)
        const DColBoost& cb = cols[Q.col_begin + k];
        bool apply = root;
        for (uint32_t sk = 0; sk < cb.nskip; ++sk)
            if (fabsf(cb.skip[sk] - score) < 0.00001f) apply = false;
        uint32_t row = 0;
        if (apply) {
            apply = doc >= cb.key_base && doc - cb.key_base < cb.num_keys;
            row = doc - cb.key_base;
        }
        if (apply && cb.present) apply = ((cb.present[row >> 5] >> (row & 31u)) & 1u) != 0u;
        float factor = 0.0f;
        if (apply) {
            const float v = cb.values[row];
            if (cb.fun == BF_LOG10) factor = log10_f32(v + cb.param);  // boost.rs:297-300: only Log10 records its factor
            score = apply_boost_value(score, cb, v);
        }
        T_cols[3u * k] = apply ? 1u : 0u;
        T_cols[3u * k + 1u] = __float_as_uint(factor);
        T_cols[3u * k + 2u] = __float_as_uint(score);
    }
    uint32_t* T_end = T_cols + 3u * Q.n_col;
    T_end[0] = root ? 1u : 0u;
    T_end[1] = __float_as_uint(tree_score);
    T_end[2] = __float_as_uint(score);
}
void launch_explain(hipStream_t st, uint32_t n_docs, const ExQuery* queries, const uint32_t* doc_query, const uint32_t* docs, const ExOp* ops, const uint16_t* aux,
                    const ExList* lists, const DColBoost* cols, uint32_t* trace) {
    if (!n_docs) return;
    hipLaunchKernelGGL(k_explain, dim3((n_docs + 63u) / 64u), dim3(64), 0, st, n_docs, queries, doc_query, docs, ops, aux, lists, cols, trace);
}

}  // namespace vq
