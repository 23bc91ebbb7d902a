// CDNA4 (gfx950) kernels of the veloci query path.  wave64 everywhere; no MFMA (integer / gather /
// scalar-f32 work, HBM bound).  Compiled with -ffp-contract=off and correctly-rounded f32 division so
// that every score is bit-identical to the reference's scalar Rust arithmetic.
//
//   k_tile_scan    K1+K2+K3+K4+K5+K6+K7+K10+K11 fused: one workgroup owns a contiguous span of the
//                  shard's doc-id space for one query and walks it tile by tile:
//                    stream the doc ids of every list that falls into the tile (16 B/lane coalesced
//                    loads) -> per-list LDS bitmaps -> postfix presence program on bitmap words
//                    (AND/OR/filter) -> for surviving docs only: rank = prefix popcount -> gather the
//                    f16 anchor scores -> reference score arithmetic -> boosts -> facet histogram ->
//                    per-workgroup exact top-k (64-bit keys, LDS candidate buffer + bitonic prune)
//   k_merge_spans  per query: merge the span-local top-k lists into the shard partial
//   k_finalize     per query: merge the partials of all shards (after the RCCL all-gather) into the
//                  final ranked hits; sum hit counts
//   k_hist_reduce  sum facet histograms of all shards
//   k_facet_select per (query, facet): top-`top` histogram entries (count desc, value id asc)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "device_types.hpp"
#include "kernels.hpp"

namespace vq {

// ------------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t y = __shfl_xor(v, o, 64);
        v = y < v ? y : v;
    }
    return v;
}

// First index in a[0..n) with a[idx] >= target (a ascending).  Whole wave cooperates: 64 probes per
// round, ~log64(n) dependent rounds instead of log2(n).
__device__ uint32_t wave_lower_bound(const uint32_t* __restrict__ a, uint32_t n, uint32_t target) {
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

// exclusive prefix popcount of `words[0..ww)` into pre[0..ww) by one wave
__device__ void wave_prefix_popc(const uint32_t* words, uint16_t* pre, uint32_t ww) {
    const uint32_t lane = lane_id();
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < ww; c0 += 64) {
        uint32_t idx = c0 + lane;
        uint32_t x = idx < ww ? (uint32_t)__popc(words[idx]) : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += y;
        }
        if (idx < ww) pre[idx] = (uint16_t)(carry + incl - x);
        carry += __shfl(incl, 63, 64);
    }
}

// ------------------------------------------------------------------------------------ candidate buffer
// LDS buffer of 64-bit keys; prune = bitonic sort (descending) + keep k + raise the threshold.
struct CandState {
    unsigned long long* cand;  // [kCandCap]
    uint32_t* n;               // pushes so far (may exceed kCandCap)
    unsigned long long* thr;   // keys <= thr cannot enter the top-k any more
};

__device__ void cand_prune(const CandState& cs, uint32_t k) {
    __syncthreads();
    uint32_t n = *cs.n;
    if (n > (uint32_t)kCandCap) n = kCandCap;
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = n + threadIdx.x; i < m; i += kBlock) cs.cand[i] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += kBlock) {
                uint32_t i = ((t / stride) * stride << 1) + (t % stride);
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
        uint32_t nn = n < k ? n : k;
        *cs.n = nn;
        if (n >= k) *cs.thr = cs.cand[k - 1];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ score arithmetic
__device__ __forceinline__ float posting_value(float term_score, uint16_t f16bits) {
    // search_field.rs:426  hit.score * (el.score.to_f32() / 100.0)
    float a = __half2float(__ushort_as_half(f16bits));
    return term_score * (a / 100.0f);
}

__device__ __forceinline__ float log10_f32(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float log2_f32(float x) { return (float)log2((double)x); }

__device__ float apply_col_boost(float score, const DColBoost& cb, uint32_t doc) {
    // add_boost boost.rs:470-504 + apply_boost :283-377
    for (uint32_t s = 0; s < cb.nskip; ++s)
        if (fabsf(cb.skip[s] - score) < 0.00001f) return score;
    if (doc < cb.key_base) return score;
    uint32_t row = doc - cb.key_base;
    if (row >= cb.num_keys) return score;
    if (cb.present && !((cb.present[row >> 5] >> (row & 31u)) & 1u)) return score;
    float v = cb.values[row];
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

// ------------------------------------------------------------------------------------ k_tile_scan
// LDS map (u32 units), fixed part first:
//   [0 .. 2*kCandCap)                        candidate keys (u64)
//   misc: thr(2) cand_n next_head hits_acc(2) pad
//   cur[kMaxLists] cnt_lo[kMaxLists] cnt_hi[kMaxLists]
//   stack[stack_depth*kBlock]               postfix stack, one column per thread (words, then scores)
//   rootw[WW]  bm[L*WW]  pre[L*WW] (u16)
constexpr uint32_t kLdsCand = 0;
constexpr uint32_t kLdsMisc = kLdsCand + 2 * kCandCap;
constexpr uint32_t kLdsCur = kLdsMisc + 8;
constexpr uint32_t kLdsCntLo = kLdsCur + kMaxLists;
constexpr uint32_t kLdsCntHi = kLdsCntLo + kMaxLists;
constexpr uint32_t kLdsStack = kLdsCntHi + kMaxLists;

size_t tile_scan_lds_bytes(uint32_t n_lists, uint32_t tile_words, uint32_t stack_depth) {
    size_t u32s = kLdsStack + (size_t)stack_depth * kBlock + (size_t)tile_words + (size_t)n_lists * tile_words + ((size_t)n_lists * tile_words + 1) / 2;
    return u32s * 4 + 16;
}

__device__ __forceinline__ uint32_t eval_presence_word(const DOp* __restrict__ ops, uint32_t n_ops, const uint32_t* bm, uint32_t ww, uint32_t w,
                                                       uint32_t* stack /* column of this thread, stride kBlock */) {
    uint32_t sp = 0;
    for (uint32_t o = 0; o < n_ops; ++o) {
        const DOp op = ops[o];
        uint32_t v;
        if (op.kind == OP_LEAF) {
            v = 0;
            for (uint32_t j = 0; j < op.list_count; ++j) v |= bm[(op.list_begin + j) * ww + w];
        } else if (op.kind == OP_AND) {
            v = 0xFFFFFFFFu;
            for (uint32_t c = 0; c < op.nchild; ++c) v &= stack[(sp - 1 - c) * kBlock];
            sp -= op.nchild;
        } else {
            v = 0;
            for (uint32_t c = 0; c < op.nchild; ++c) v |= stack[(sp - 1 - c) * kBlock];
            sp -= op.nchild;
        }
        stack[sp * kBlock] = v;
        ++sp;
    }
    return sp ? stack[0] : 0u;
}

__global__ __launch_bounds__(kBlock) void k_tile_scan(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                      const uint32_t* __restrict__ span_base, uint32_t nq, uint32_t stack_depth,
                                                      unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits,
                                                      uint32_t* __restrict__ hist) {
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
    const uint32_t span = blockIdx.x - span_base[q];
    const uint8_t* blob = blobs + blob_off[q];
    const QHeader* H = reinterpret_cast<const QHeader*>(blob);
    const uint32_t L = H->n_lists;
    const uint32_t WW = H->tile_words;
    const uint32_t W = WW << 5;
    const uint32_t top_k = H->top_k;
    const DList* __restrict__ lists = reinterpret_cast<const DList*>(blob + H->off_lists);
    const DOp* __restrict__ ops = reinterpret_cast<const DOp*>(blob + H->off_ops);
    const DOp* __restrict__ fops = reinterpret_cast<const DOp*>(blob + H->off_fops);
    const DGroup* __restrict__ groups = reinterpret_cast<const DGroup*>(blob + H->off_groups);
    const DTermBoost* __restrict__ tboosts = reinterpret_cast<const DTermBoost*>(blob + H->off_tboost);
    const DColBoost* __restrict__ cols = reinterpret_cast<const DColBoost*>(blob + H->off_col);
    const DLocField* __restrict__ locf = reinterpret_cast<const DLocField*>(blob + H->off_locf);
    const DFacet* __restrict__ facets = reinterpret_cast<const DFacet*>(blob + H->off_facets);
    const uint32_t n_ops = H->n_ops, n_fops = H->n_fops;
    const uint32_t n_groups = H->n_groups, n_tboost = H->n_tboost, n_col = H->n_col, n_locf = H->n_locf, n_facets = H->n_facets;

    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + kLdsCand);
    uint32_t* stack = lds + kLdsStack + tid;  // this thread's column
    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds + kLdsMisc);
    uint32_t* cand_n = lds + kLdsMisc + 2;
    uint32_t* next_head = lds + kLdsMisc + 3;
    uint32_t* hits_acc = lds + kLdsMisc + 4;
    uint32_t* cur = lds + kLdsCur;
    uint32_t* cnt_lo = lds + kLdsCntLo;
    uint32_t* cnt_hi = lds + kLdsCntHi;
    uint32_t* rootw = lds + kLdsStack + stack_depth * kBlock;
    uint32_t* bm = rootw + WW;
    uint16_t* pre = reinterpret_cast<uint16_t*>(bm + L * WW);
    CandState cs{cand, cand_n, thr};

    // ---- span of the doc-id space owned by this workgroup
    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const uint32_t span_lo = H->doc_lo + (uint32_t)(range * span / n_spans);
    const uint32_t span_hi = H->doc_lo + (uint32_t)(range * (span + 1) / n_spans);

    // ---- initial cursors: first entry >= span_lo of every list
    for (uint32_t i = wave; i < L; i += kBlock / 64) {
        uint32_t c = wave_lower_bound(lists[i].docs, lists[i].len, span_lo);
        if (lane == 0) cur[i] = c;
    }
    if (tid == 0) {
        *thr = 0ull;
        *cand_n = 0;
        *hits_acc = 0;
    }
    uint32_t my_hits = 0;
    __syncthreads();

    while (true) {
        // ---- P0: next tile = the tile holding the smallest pending doc of the cover lists
        if (wave == 0) {
            uint32_t h = 0xFFFFFFFFu;
            for (uint32_t i = lane; i < L; i += 64) {
                if (lists[i].flags & LIST_COVER) {
                    uint32_t c = cur[i];
                    if (c < lists[i].len) {
                        uint32_t d = lists[i].docs[c];
                        h = d < h ? d : h;
                    }
                }
            }
            h = wave_min_u32(h);
            if (lane == 0) *next_head = h;
        }
        __syncthreads();
        const uint32_t head = *next_head;
        if (head >= span_hi) break;  // uniform: span exhausted
        const uint32_t tile_lo = head & ~(W - 1u);
        const uint32_t tile_end = tile_lo + W;  // may wrap to 0 at the top of the id space
        const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;

        // ---- P0b: lists outside the cover may be far behind: skip ahead with a wave-wide search
        for (uint32_t i = wave; i < L; i += kBlock / 64) {
            const uint32_t c = cur[i];
            const uint32_t len = lists[i].len;
            if (c + 256u < len && lists[i].docs[c + 256u] < tile_lo) {  // uniform per wave
                uint32_t adv = wave_lower_bound(lists[i].docs + c, len - c, tile_lo);
                if (lane == 0) cur[i] = c + adv;
            }
        }
        // ---- P1: clear the tile state
        for (uint32_t x = tid; x < L * WW; x += kBlock) bm[x] = 0u;
        if (tid < L) {
            cnt_lo[tid] = 0u;
            cnt_hi[tid] = 0u;
        }
        __syncthreads();

        // ---- P2: stream every list's doc ids of this tile into its bitmap (16 B per lane)
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t c0 = cur[i];
            const uint32_t len = lists[i].len;
            const uint4* __restrict__ docs4 = reinterpret_cast<const uint4*>(lists[i].docs);
            const uint32_t nvec = (len + 3u) >> 2;
            uint32_t nlo = 0, nhi = 0;
            uint32_t* bmi = bm + i * WW;
            for (uint32_t v = (c0 >> 2) + tid; v < nvec; v += kBlock) {
                const uint4 d4 = docs4[v];
                const uint32_t base = v << 2;
                const uint32_t dd[4] = {d4.x, d4.y, d4.z, d4.w};
                bool stop = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t idx = base + j;
                    const uint32_t d = dd[j];
                    if (idx < c0) continue;
                    if (idx >= len || d >= tile_hi) {
                        stop = true;
                        break;
                    }
                    ++nhi;
                    if (d < tile_lo) {
                        ++nlo;
                        continue;
                    }
                    const uint32_t rel = d - tile_lo;
                    atomicOr(&bmi[rel >> 5], 1u << (rel & 31u));
                }
                if (stop) break;
            }
            if (nhi) atomicAdd(&cnt_hi[i], nhi);
            if (nlo) atomicAdd(&cnt_lo[i], nlo);
        }
        __syncthreads();

        // ---- P3: presence program on bitmap words -> root words
        uint32_t surv = 0;
        for (uint32_t w = tid; w < WW; w += kBlock) {
            uint32_t r = eval_presence_word(ops, n_ops, bm, WW, w, stack);
            if (n_fops) r &= eval_presence_word(fops, n_fops, bm, WW, w, stack);
            // docs of the tile that lie outside [span_lo, span_hi) can not be set: cursors start at
            // span_lo and the scatter stops at tile_hi <= span_hi
            rootw[w] = r;
            surv += (uint32_t)__popc(r);
        }
        const int any = __syncthreads_or(surv != 0);

        if (any) {
            // ---- P4: rank support for score gathers
            for (uint32_t i = wave; i < L; i += kBlock / 64)
                if (lists[i].flags & LIST_HAS_SCORES) wave_prefix_popc(bm + i * WW, pre + i * WW, WW);
            __syncthreads();

            // ---- P5: score the surviving docs, run the sink stages, feed the top-k
            uint32_t w = tid;
            uint32_t r = w < WW ? rootw[w] : 0u;
            bool pending = false;
            unsigned long long pend_key = 0ull;
            float* fstack = reinterpret_cast<float*>(stack);
            while (true) {
                while (true) {
                    if (pending) {
                        if (pend_key > *thr) {
                            uint32_t pos = atomicAdd(cand_n, 1u);
                            if (pos < (uint32_t)kCandCap) {
                                cand[pos] = pend_key;
                                pending = false;
                            } else break;
                        } else pending = false;
                    }
                    if (r == 0u) {
                        w += kBlock;
                        if (w >= WW) break;
                        r = rootw[w];
                        continue;
                    }
                    const uint32_t b = (uint32_t)__ffs((int)r) - 1u;
                    r &= r - 1u;
                    const uint32_t doc = tile_lo + (w << 5) + b;
                    const uint32_t below = (1u << b) - 1u;

                    // -- score tree (postfix), stack slot = uniform program counter state
                    uint32_t sp = 0;
                    uint32_t pmask = 0;  // bit s: stack slot s holds a present value
                    for (uint32_t o = 0; o < n_ops; ++o) {
                        const DOp op = ops[o];
                        float s = 0.0f;
                        bool present = false;
                        if (op.kind == OP_LEAF) {
                            for (uint32_t j = 0; j < op.list_count; ++j) {
                                const uint32_t li = op.list_begin + j;
                                const uint32_t word = bm[li * WW + w];
                                if ((word >> b) & 1u) {
                                    float v = 0.0f;
                                    if (lists[li].flags & LIST_HAS_SCORES) {
                                        const uint32_t rank = (uint32_t)pre[li * WW + w] + (uint32_t)__popc(word & below);
                                        const uint32_t idx = cur[li] + cnt_lo[li] + rank;
                                        v = posting_value(lists[li].term_score, lists[li].scores[idx]);
                                    }
                                    if (!present || v > s) s = v;  // dedup keeps the max (search_field.rs:455-461)
                                    present = true;
                                }
                            }
                        } else if (op.kind == OP_AND) {
                            const uint32_t base = sp - op.nchild;
                            present = true;
                            for (uint32_t c = 0; c < op.nchild; ++c) present = present && ((pmask >> (base + c)) & 1u);
                            if (present) {
                                s = 0.0f;  // set_op.rs:415-416: others summed first, the shortest list's score last
                                for (uint32_t c = 0; c < op.nchild; ++c) s += fstack[(base + op.and_order[c]) * kBlock];
                            }
                            sp = base;
                        } else {
                            const uint32_t base = sp - op.nchild;
                            float sum = 0.0f;
                            float nd = 0.0f;
                            for (uint32_t slot = 0; slot < op.nslots; ++slot) {  // set_op.rs:169-186
                                float m = 0.0f;
                                for (uint32_t c = 0; c < op.nchild; ++c) {
                                    if (op.child_slot[c] == slot && ((pmask >> (base + c)) & 1u)) {
                                        present = true;
                                        m = fmaxf(m, fstack[(base + c) * kBlock]);
                                    }
                                }
                                if (m >= 0.00001f) nd += 1.0f;
                                sum += m;
                            }
                            s = sum * nd * nd;
                            sp = base;
                        }
                        fstack[sp * kBlock] = s;
                        pmask = present ? (pmask | (1u << sp)) : (pmask & ~(1u << sp));
                        ++sp;
                    }
                    float score = fstack[0];

                    // -- sink stages in the reference's order
                    for (uint32_t c = 0; c < n_col; ++c) score = apply_col_boost(score, cols[c], doc);
                    for (uint32_t g = 0; g < n_groups; ++g) {
                        bool in = false;
                        for (uint32_t j = 0; j < groups[g].list_count; ++j) in = in || ((bm[(groups[g].list_begin + j) * WW + w] >> b) & 1u);
                        if (in) score *= groups[g].mult;
                    }
                    for (uint32_t t = 0; t < n_tboost; ++t)
                        if ((bm[tboosts[t].list * WW + w] >> b) & 1u) score *= tboosts[t].mult;
                    if (n_locf) {  // boost.rs:11-87: 2*c*c per field with c > 1, the MINIMUM over fields (:25)
                        float best = 0.0f;
                        bool have = false;
                        for (uint32_t f = 0; f < n_locf; ++f) {
                            uint32_t c = 0;
                            for (uint32_t j = 0; j < locf[f].list_count; ++j) c += (bm[(locf[f].list_begin + j) * WW + w] >> b) & 1u;
                            if (c > 1u) {
                                float bv = 2.0f * (float)c * (float)c;
                                if (!have || bv < best) best = bv;
                                have = true;
                            }
                        }
                        if (have) score *= best;
                    }
                    for (uint32_t f = 0; f < n_facets; ++f) {  // persistence.rs:164-175 count_values_for_ids
                        const DFacet& fa = facets[f];
                        if (doc >= fa.key_base && doc - fa.key_base < fa.num_keys) {
                            const uint32_t row = doc - fa.key_base;
                            const unsigned long long e0 = fa.offsets[row], e1 = fa.offsets[row + 1];
                            for (unsigned long long e = e0; e < e1; ++e) {
                                const uint32_t v = fa.values[e];
                                if (v < fa.num_values) atomicAdd(&hist[fa.hist_off + v], 1u);
                            }
                        }
                    }
                    ++my_hits;
                    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
                    if (key > *thr) {
                        uint32_t pos = atomicAdd(cand_n, 1u);
                        if (pos < (uint32_t)kCandCap) cand[pos] = key;
                        else {
                            pending = true;
                            pend_key = key;
                            break;
                        }
                    }
                }
                const int need = __syncthreads_or(pending ? 1 : 0);
                if (!need) break;
                cand_prune(cs, top_k);
            }
        }
        // ---- P6: advance the cursors past this tile
        __syncthreads();
        if (tid < L) cur[tid] += cnt_hi[tid];
        // cur[] is next read by wave 0 (P0) — same wave as the writers (L <= 64), LDS is in order
    }

    // ---- span done: publish the local top-k and the hit count
    cand_prune(cs, top_k);
    {
        const uint32_t n = *cand_n;
        unsigned long long* out = span_keys + (size_t)H->keys_base + (size_t)span * top_k;
        for (uint32_t i = tid; i < top_k; i += kBlock) out[i] = i < n ? cand[i] : 0ull;
    }
    if (my_hits) atomicAdd(hits_acc, my_hits);
    __syncthreads();
    if (tid == 0 && *hits_acc) atomicAdd(&num_hits[q], (unsigned long long)*hits_acc);
}

// ------------------------------------------------------------------------------------ merges
__global__ __launch_bounds__(kBlock) void k_merge_spans(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                        const unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ part_keys) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ uint32_t misc[4];
    const QHeader* H = reinterpret_cast<const QHeader*>(blobs + blob_off[blockIdx.x]);
    const uint32_t top_k = H->top_k;
    const unsigned long long* src = span_keys + H->keys_base;
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc)};
    if (threadIdx.x == 0) {
        *cs.thr = 0ull;
        *cs.n = 0;
    }
    __syncthreads();
    for (uint32_t s = 0; s < H->n_spans; ++s) {
        if (*cs.n + top_k > (uint32_t)kCandCap) cand_prune(cs, top_k);
        const uint32_t base = *cs.n;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) cand[base + i] = src[(size_t)s * top_k + i];
        __syncthreads();
        if (threadIdx.x == 0) *cs.n = base + top_k;
        __syncthreads();
    }
    cand_prune(cs, top_k);
    unsigned long long* out = part_keys + H->part_keys_off;
    for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) out[i] = i < *cs.n ? cand[i] : 0ull;
}

// gathered: num_shards packed partial buffers, shard-major, each `part_bytes` long.
__global__ __launch_bounds__(kBlock) void k_finalize(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                     const uint8_t* __restrict__ gathered, uint32_t num_shards, PartialLayout lay,
                                                     uint32_t* __restrict__ res_ids, float* __restrict__ res_scores, uint32_t* __restrict__ res_n,
                                                     unsigned long long* __restrict__ res_hits) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ uint32_t misc[4];
    const uint32_t q = blockIdx.x;
    const QHeader* H = reinterpret_cast<const QHeader*>(blobs + blob_off[q]);
    const uint32_t top_k = H->top_k;
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc)};
    if (threadIdx.x == 0) {
        *cs.thr = 0ull;
        *cs.n = 0;
    }
    __syncthreads();
    unsigned long long hits = 0;
    for (uint32_t s = 0; s < num_shards; ++s) {
        const uint8_t* pb = gathered + (size_t)s * lay.bytes;
        const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(pb + lay.off_keys) + H->part_keys_off;
        hits += reinterpret_cast<const unsigned long long*>(pb + lay.off_hits)[q];
        if (*cs.n + top_k > (uint32_t)kCandCap) cand_prune(cs, top_k);
        const uint32_t base = *cs.n;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) cand[base + i] = keys[i];
        __syncthreads();
        if (threadIdx.x == 0) *cs.n = base + top_k;
        __syncthreads();
    }
    cand_prune(cs, top_k);
    const uint32_t n = *cs.n;
    uint32_t real = 0;
    for (uint32_t i = threadIdx.x; i < top_k; i += kBlock) {
        const unsigned long long k = i < n ? cand[i] : 0ull;
        res_ids[H->part_keys_off + i] = (uint32_t)(k & 0xFFFFFFFFull);
        res_scores[H->part_keys_off + i] = __uint_as_float(unorder_f32((uint32_t)(k >> 32)));
        if (k != 0ull) ++real;
    }
    // number of real hits in the top-k window = min(total hits, top_k): keys are unique and non-zero
    if (threadIdx.x == 0) {
        res_hits[q] = hits;
        res_n[q] = hits < (unsigned long long)top_k ? (uint32_t)hits : top_k;
    }
    (void)real;
}

__global__ void k_hist_reduce(const uint8_t* __restrict__ gathered, uint32_t num_shards, PartialLayout lay, uint32_t* __restrict__ out) {
    const size_t n = lay.total_hist;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t s = 0;
        for (uint32_t p = 0; p < num_shards; ++p) s += reinterpret_cast<const uint32_t*>(gathered + (size_t)p * lay.bytes + lay.off_hist)[i];
        out[i] = s;
    }
}

// one workgroup per facet entry of the batch
__global__ __launch_bounds__(kBlock) void k_facet_select(const FacetJob* __restrict__ jobs, const uint32_t* __restrict__ hist,
                                                         uint32_t* __restrict__ out_vals, uint32_t* __restrict__ out_counts,
                                                         uint32_t* __restrict__ out_n) {
    __shared__ unsigned long long cand[kCandCap];
    __shared__ uint32_t misc[4];
    const FacetJob job = jobs[blockIdx.x];
    CandState cs{cand, misc + 2, reinterpret_cast<unsigned long long*>(misc)};
    if (threadIdx.x == 0) {
        *cs.thr = 0ull;
        *cs.n = 0;
    }
    __syncthreads();
    const uint32_t* h = hist + job.hist_off;
    const uint32_t k = job.top;
    for (uint32_t base = 0; base < job.num_values; base += kBlock) {  // uniform trip count
        const uint32_t v = base + threadIdx.x;
        bool pending = false;
        unsigned long long key = 0ull;
        if (v < job.num_values) {
            const uint32_t c = h[v];
            if (c) {
                key = ((unsigned long long)c << 32) | (unsigned long long)(0xFFFFFFFFu - v);  // count desc, value id asc
                pending = key > *cs.thr;
            }
        }
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
    cand_prune(cs, k);
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
                      uint32_t nq, uint32_t stack_depth, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist) {
    if (!total_spans) return;
    hipLaunchKernelGGL(k_tile_scan, dim3(total_spans), dim3(kBlock), lds_bytes, st, blobs, blob_off, span_base, nq, stack_depth, span_keys, num_hits, hist);
}
void launch_merge_spans(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const unsigned long long* span_keys,
                        unsigned long long* part_keys) {
    if (!nq) return;
    hipLaunchKernelGGL(k_merge_spans, dim3(nq), dim3(kBlock), 0, st, blobs, blob_off, span_keys, part_keys);
}
void launch_finalize(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const uint8_t* gathered, uint32_t num_shards,
                     const PartialLayout& lay, uint32_t* res_ids, float* res_scores, uint32_t* res_n, unsigned long long* res_hits) {
    if (!nq) return;
    hipLaunchKernelGGL(k_finalize, dim3(nq), dim3(kBlock), 0, st, blobs, blob_off, gathered, num_shards, lay, res_ids, res_scores, res_n, res_hits);
}
void launch_hist_reduce(hipStream_t st, const uint8_t* gathered, uint32_t num_shards, const PartialLayout& lay, uint32_t* out) {
    if (!lay.total_hist) return;
    uint32_t blocks = (uint32_t)((lay.total_hist + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_hist_reduce, dim3(blocks), dim3(256), 0, st, gathered, num_shards, lay, out);
}
void launch_facet_select(hipStream_t st, uint32_t n_jobs, const FacetJob* jobs, const uint32_t* hist, uint32_t* out_vals, uint32_t* out_counts,
                         uint32_t* out_n) {
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_facet_select, dim3(n_jobs), dim3(kBlock), 0, st, jobs, hist, out_vals, out_counts, out_n);
}

}  // namespace vq
