// k_scan_probe — AND of 2..4 single-list posting leaves whose cover (the sparsest operand) has a tile-packed image and whose other
// operands are read per tile either as bitmap words (dense lists) or as sorted 16-bit arrays (lists below 1/16 of the docs: 2 B per
// posting instead of a bit per doc — Roaring's two containers): the headline shape (3-term AND, df 10 % / 3 % / 1 % of the docs) reads
// the 10 % list as words, the 3 % list as arrays and streams the 1 % list.
//
// The doc space is walked in tiles of 32768 docs.  The cover's tile-packed image (DProbeLeaf::cov32, one word per posting: in-tile doc
// offset << 16 | f16 score, every tile padded to 8 entries; its directory `gdir` says where a tile starts) is streamed, one 16 B/lane
// load per 256 postings — no search, no counting, every lane slot holds a posting of the tile or an all-ones pad.  The operands' tile —
// 1024 bitmap words, or up to 2048 16-bit offsets — sits in LDS (coalesced 16 B/lane loads; an array's loads are cut to its entries).
// Every cover posting tests its bit in the bitmap operands' words; with array operands, the postings that passed are then looked up in
// the arrays by a binary search in LDS, 64 at a time.  Nothing is computed per bitmap word or per array entry: the work per tile
// follows the COVER's postings (about 330 per tile in the headline query).
//
// Loads are only ISSUED at the top of a tile — the next tile's words, rank entries and cover postings, the score gathers of the flush
// in flight, the query's shared threshold — and only consumed behind the top of the next one, which waits for everything in flight
// once: nothing in the loop waits for a load it has just issued, and a whole tile of work hides the latency.
//
// A doc that is in every operand is a hit.  Its score is the AND's ordered sum (set_op.rs:415-416) of the posting values
// s_t * (f16 / 100) (search_field.rs:426).  The top-k only needs the best few, so a hit is scored only if it can still enter:
// the sum with the cover's value known and every other operand at its list maximum (DList::max_raw) bounds the score from
// above — f32 add and mul are monotone — and that bound is monotone in the cover's raw f16 score, so the test per posting is one
// integer compare against `raw_min`, recomputed whenever the threshold moves.  Pruned hits are still counted (num_hits is exact).
//
// Hits that stay live need their index in every operand (the f16 score is scores[index]): in a bitmap operand the rank directory entry
// of the doc's 512-doc group (staged in LDS with the tile) + popcount of the group's words below the doc; in an array operand the
// position the search found (the score is the low half of cov32 at the same position).  Ranked hits are queued and scored 64 at a
// time, all score gathers of a flush in flight together.
// Same results as k_scan_simple / k_tile_scan bit for bit (tests run every such query through all three).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "device_types.hpp"
#include "kernel_common.hpp"
#include "kernels.hpp"
#include "probe_common.hpp"

namespace vq {

// Diagnostic build only (make stamp): time shares of the kernel's phases, summed over all waves (s_memtime ticks), and event counts.
#ifdef VQ_STAMP
__device__ unsigned long long g_probe_stamp[16];
#define PS_INIT                                                      \
    unsigned long long _st0 = __builtin_amdgcn_s_memtime();          \
    const unsigned long long _t0 = _st0;                             \
    const unsigned long long _r0 = __builtin_amdgcn_s_memrealtime(); \
    unsigned long long _acc[16] = {0};
#define PS_AT(k)                                                \
    {                                                           \
        unsigned long long _st1 = __builtin_amdgcn_s_memtime(); \
        _acc[k] += _st1 - _st0;                                 \
        _st0 = _st1;                                            \
    }
#define PS_COUNT(k) _acc[k] += 1ull;
#define PS_FLUSH                                                                                             \
    _acc[14] = __builtin_amdgcn_s_memtime() - _t0;     /* shader-clock ticks of the span */                   \
    _acc[15] = __builtin_amdgcn_s_memrealtime() - _r0; /* constant 100 MHz ticks of the span */               \
    if (threadIdx.x == 0) {                                                                                  \
        _Pragma("unroll") for (int _k = 0; _k < 16; ++_k) if (_acc[_k]) atomicAdd(&g_probe_stamp[_k], _acc[_k]); \
    }
void debug_read_probe_stamps(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_stamp), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_stamp), z, sizeof z);
    }
}
#else
#define PS_INIT
#define PS_AT(k)
#define PS_COUNT(k)
#define PS_FLUSH
#endif

constexpr uint32_t kPMaxR = 2;                  // rounds of 256 cover postings of a tile that are prefetched into registers (more: fetched on the spot)
constexpr uint32_t kPU = 64 + kPMaxR * 256;     // unranked queue: live hits of the current tile, (doc - tile_lo) << 16 | raw f16 score of the cover
constexpr uint32_t kPR = 128;                   // ranked queue
// LDS map (u32): misc[8] | shape[32] | uq[kPU] | rdoc[kPR] rraw[kPR] ridx[ND][kPR] | tile[ND][kPTW] | rank[ND][kPRk] | cand[2 * cand_cap]
// (the candidate buffer, the only part sized at run time, comes last: every other offset is a constant of the instantiation)
constexpr uint32_t kPLdsShape = 8;
constexpr uint32_t kPLdsU = kPLdsShape + 32;
constexpr uint32_t kPLdsR = kPLdsU + kPU;
__host__ __device__ constexpr uint32_t probe_lds_tile(uint32_t nd) { return kPLdsR + (2 + nd) * kPR; }
__host__ __device__ constexpr uint32_t probe_lds_cand(uint32_t nd) { return probe_lds_tile(nd) + nd * (kPTW + kPRk); }
size_t scan_probe_lds_bytes(uint32_t cand_cap, uint32_t nd) { return (size_t)(probe_lds_cand(nd) + 2 * cand_cap) * 4 + 16; }

// OR: the root is an OR of leaves with a term slot each (set_op.rs:87-220) — see the kernel's comment
template <uint32_t ND, uint32_t NA, bool OR = false>  // ND operands beside the cover, the last NA of them (roles NB .. ND-1) probed as 16-bit arrays
__device__ __forceinline__ void probe_body(const uint8_t* __restrict__ blob, const uint32_t span, const uint32_t q, const uint32_t cand_cap,
                                           unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const VQ_CONST QHeader* H = as_const<QHeader>(blob);
    const VQ_CONST DList* gl = as_const<DList>(blob + H->off_lists);
    const VQ_CONST DOp* gops = as_const<DOp>(blob + H->off_ops);
    const VQ_CONST DProbe* P = as_const<DProbe>(blob + H->off_simple2);
    const uint32_t sflags = H->simple_flags;
    const uint32_t top_k = H->top_k;
    constexpr uint32_t n = ND + 1u;
    constexpr uint32_t NB = ND - NA;             // bitmap operands: roles 0 .. NB-1
    constexpr uint32_t NB1 = NB ? NB : 1u, NA1 = NA ? NA : 1u;  // (array extents)
    PS_INIT

    unsigned long long* thr = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* cand_n = lds + 2;
    uint32_t* sh = lds + kPLdsShape;
    uint32_t* uq = lds + kPLdsU;
    uint32_t* rq = lds + kPLdsR;  // rdoc[kPR] rraw[kPR] ridx[ND][kPR]
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(lds + probe_lds_cand(ND));
    uint32_t* tile = lds + probe_lds_tile(ND);  // [ND][kPTW]: a bitmap operand's 1024 words / an array operand's (up to) 2048 16-bit offsets
    uint32_t* rank = tile + ND * kPTW;          // [ND][kPRk] (bitmap operands only)
    unsigned long long* const gthr = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blob) + offsetof(QHeader, gthr));
    CandState cs{cand, cand_n, thr, cand_cap, gthr};
    cs.upper = H->key_upper;
    uint32_t* const stat = H->stat_off ? lds + 4 : nullptr;
    uint8_t* const pool = H->off_pool ? const_cast<uint8_t*>(blob) + H->off_pool : nullptr;

    // ---- the query's shape: the cover leaf streams its tile-packed postings, the others are read as bitmap words or 16-bit arrays
    const uint32_t* ccov = nullptr;
    const uint32_t* cgdir = nullptr;
    const uint32_t* d_bitmap[NB1];
    const uint32_t* d_rank[NB1];
    const uint16_t* d_arr[NA1];
    const uint32_t* d_agd[NA1];
    {
        const uint32_t ck = (uint32_t)__ffs((int)((sflags >> 8) & 0xFu)) - 1u;
        const uint32_t am = (sflags >> 12) & 0xFu;  // leaf k is an array operand (the host instantiates NA = their number)
        uint32_t role_of[4] = {0u, 0u, 0u, 0u};     // leaf k -> 0 = cover, 1 + i = operand i
        uint32_t ib = 0, ia = NB;
#pragma unroll
        for (uint32_t k = 0; k < n; ++k) {
            const VQ_CONST DList& d = gl[gops[k].list_begin];
            if (k == ck) {
                ccov = P->leaf[k].cov32;
                cgdir = P->leaf[k].gdir;
                if (lane == 0) {
                    sh[kShCts] = __float_as_uint(d.term_score);
                    sh[kShPrunable] = (d.term_score > 0.0f && d.max_raw < 0x7C00u) ? 1u : 0u;
                }
            } else {
                const bool is_arr = ((am >> k) & 1u) != 0u;  // uniform
                const uint32_t role = is_arr ? ia : ib;
#pragma unroll
                for (uint32_t j = 0; j < NB; ++j)
                    if (!is_arr && j == role) {
                        d_bitmap[j] = d.bitmap;
                        d_rank[j] = d.rank_dir;
                    }
#pragma unroll
                for (uint32_t a = 0; a < NA; ++a)
                    if (is_arr && NB + a == role) {
                        d_arr[a] = P->leaf[k].arr16;
                        d_agd[a] = P->leaf[k].gdir;
                    }
                if (lane == 0) {
                    const uint16_t mr = d.max_raw;
                    sh[kShTs + role] = __float_as_uint(d.term_score);
                    sh[kShVmax + role] = (d.term_score > 0.0f && mr < 0x7C00u) ? __float_as_uint(posting_value(d.term_score, mr)) : 0x7F800000u;  // +inf: no bound
                    // where the operand's f16 scores are gathered from, as a u16 array: the list's own scores by posting index, or the low
                    // halves of its tile-packed words by (2 x) packed position
                    reinterpret_cast<unsigned long long*>(sh + kShScores)[role] = is_arr ? (unsigned long long)(uintptr_t)P->leaf[k].cov32 : (unsigned long long)(uintptr_t)d.scores;
                }
                role_of[k] = 1u + role;
                if (is_arr) ++ia;
                else ++ib;
            }
        }
        const KOp root(gops + n);
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t k = OR ? j : root.and_order(j);  // (an OR's leaves arrive in slot order: compile.cpp sorts them)
            if (lane == 0) sh[kShSrc + j] = k == 0u ? role_of[0] : k == 1u ? role_of[1] : k == 2u ? role_of[2] : role_of[3];
        }
    }

    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & ~(kPT - 1u));
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & ~(kPT - 1u));
    const uint32_t bitmap_base = H->bitmap_base;
    const uint32_t keys_base = H->keys_base;

    if (lane == 0) {
        *thr = __hip_atomic_load(gthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // what other spans of the query have already reached
        *cand_n = 0;
        lds[4] = 0u;
    }
    __syncthreads();
    unsigned long long thr_seen = *thr;
    // the smallest raw cover score that can still reach the threshold: one for an AND, one per set of present operands for an OR
    uint32_t raw_min = 0u;
    constexpr uint32_t kMasks = OR ? (1u << ND) : 1u;
    uint32_t rmin[kMasks];
    auto take_raw_min = [&](const float thr_f) {
        if (OR) {
#pragma unroll
            for (uint32_t m = 0; m < kMasks; ++m) rmin[m] = probe_raw_min<ND>(sh, thr_f, lane, m);
        } else raw_min = probe_raw_min<ND>(sh, thr_f, lane);
    };
    take_raw_min(__uint_as_float(unorder_f32((uint32_t)(thr_seen >> 32))));

    const u32x4 kSent = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const VQ_GLOBAL u32x4* cc4 = as_global(reinterpret_cast<const u32x4*>(ccov));

    uint32_t un = 0, rn = 0;
    uint32_t ucnt = 0;  // OR: this lane's share of the set bits of the operands' ORed words
    unsigned long long hits = 0;
    unsigned long long g_prev = 0ull;

    // ---- tiles of the span; a slice of every tile-packed list's directory rides in a register (lane l: granules below tile dir_base + l)
    const uint32_t t_first = (span_lo - bitmap_base) >> kProbeTileShift;
    const uint32_t t_end = span_hi > span_lo ? ((span_hi - 1u - bitmap_base) >> kProbeTileShift) + 1u : t_first;  // one behind the last tile
    uint32_t dir_base = t_first;
    auto load_dir = [&](const uint32_t* g) { return as_global(g)[dir_base + lane < t_end ? dir_base + lane : t_end]; };  // (entry t_end exists: one behind the last tile)
    uint32_t dirv = load_dir(cgdir);
    uint32_t adv[NA1];
#pragma unroll
    for (uint32_t a = 0; a < NA; ++a) adv[a] = load_dir(d_agd[a]);
    auto dir_at = [&](const uint32_t v, const uint32_t tt) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(tt - dir_base)); };

    // registers of the tile in flight: its words / array vectors and rank entries, its cover postings (the first kPMaxR rounds)
    u32x4 wk[ND][kPNV];
    uint32_t rk[NB1];
    u32x4 nid[kPMaxR], cid[kPMaxR];
#pragma unroll
    for (uint32_t r = 0; r < kPMaxR; ++r) nid[r] = cid[r] = kSent;
    uint32_t pf_rounds = 0, pf_v0 = 0, pf_v1 = 0;
    uint32_t na_g0[NA1], na_nv[NA1];  // array operands, tile in flight: first granule, granules (= 16-byte vectors of 8 offsets)
    uint32_t a_g0[NA1], a_cnt[NA1];   // ... current tile: first granule, padded entries
#pragma unroll
    for (uint32_t a = 0; a < NA1; ++a) na_g0[a] = na_nv[a] = a_g0[a] = a_cnt[a] = 0u;
    auto issue_tile = [&](const uint32_t tt, const uint32_t g0, const uint32_t g1) {  // uniform; tt < t_end; [g0, g1): the cover's granules of tile tt; na_g0 / na_nv set
        pf_rounds = (g1 - g0 + 31u) >> 5;  // rounds of 256 postings
        pf_v0 = g0 * 2u;
        pf_v1 = g1 * 2u;
        if (OR || pf_rounds) {  // uniform: (AND) a tile without cover postings has no hits — nothing of it is read
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) {
                const VQ_GLOBAL u32x4* gb = as_global(reinterpret_cast<const u32x4*>(d_bitmap[i] + (size_t)tt * kPTW));
#pragma unroll
                for (uint32_t h = 0; h < kPNV; ++h) wk[i][h] = gb[h * 64u + lane];
                rk[i] = as_global(d_rank[i])[tt * kPRk + lane];
            }
#pragma unroll
            for (uint32_t a = 0; a < NA; ++a) {
                const VQ_GLOBAL u32x4* ga = as_global(reinterpret_cast<const u32x4*>(d_arr[a]) + na_g0[a]);
#pragma unroll
                for (uint32_t h = 0; h < kPNV; ++h)
                    if (h * 64u + lane < na_nv[a]) wk[NB + a][h] = ga[h * 64u + lane];  // (only the lanes that hold entries of the tile)
            }
#pragma unroll
            for (uint32_t r = 0; r < kPMaxR; ++r) {
                const uint32_t v = pf_v0 + r * 64u + lane;
                nid[r] = kSent;
                if (r < pf_rounds && v < pf_v1) nid[r] = cc4[v];  // (only the lanes that hold postings of the tile)
            }
        }
    };

    uint32_t tile_lo = 0;
    // ---- scoring of the ranked queue, 64 hits at a time, WITHOUT waiting for its gathers: a flush is a little pipeline that advances one
    // stage per tile.  Stage i has the gather of dense operand i's scores in flight; when it is consumed — a tile later, the data has
    // long arrived — the bound is taken again with the value now known, hits that can no longer reach the threshold drop out, and the
    // next operand's gather is issued.  After the last operand the scores are final and the keys go to the candidate buffer.
    uint32_t f_stage = 0;  // 0: idle; i + 1: operand i's gather is in flight
    uint32_t f_doc = 0, f_idx[ND];
    float f_vc = 0.0f, f_vd[ND];
    uint16_t f_r = 0;
    bool f_alive = false;
#pragma unroll
    for (uint32_t i = 0; i < ND; ++i) {
        f_idx[i] = 0u;
        f_vd[i] = 0.0f;
    }
    uint32_t rhead = 0;              // the ranked queue is a ring: entries [rhead, rhead + rn) mod kPR
    uint32_t tiles_since_merge = 8;  // pool merges are spaced out: each is a round trip to memory under a lock
    // keys of scored hits -> the candidate buffer (prune when it is full, merge into the query's pool now and then, take the new threshold)
    auto push_keys = [&](const unsigned long long key, bool pending, const bool final) {
        if (wballot(pending)) {  // uniform; rare once the threshold has warmed up
            while (true) {
                if (pending) {
                    if (key > *thr) {
                        uint32_t pos = atomicAdd(cs.n, 1u);
                        if (pos < cs.cap) {
                            cs.cand[pos] = key;
                            pending = false;
                        }
                    } else pending = false;
                }
                probe_lds_fence();
                if (!wballot(pending)) break;  // (one wave per workgroup: a ballot is the workgroup's vote)
                cand_prune(cs, top_k);
            }
            if (pool && (tiles_since_merge >= 8u || final)) {  // uniform
                PS_COUNT(13)
                tiles_since_merge = 0;
                probe_pool_merge<true>(cs, top_k, pool, lane);
            }
            const unsigned long long tn = *thr;
            if ((uint32_t)(tn >> 32) != (uint32_t)(thr_seen >> 32)) take_raw_min(__uint_as_float(unorder_f32((uint32_t)(tn >> 32))));
            thr_seen = tn;
        }
    };
    // the last stage of a flush: the scores are final, the keys go to the candidate buffer
    uint32_t f_mask = 0;  // OR: the operands that hold the doc (an absent one has no index: all ones in the ranked queue)
    auto sum_of = [&](const ProbeShape<ND>& S, const float vc, const float (&vd)[ND], const uint32_t mask) { return OR ? probe_or_sum<ND>(S, vc, vd, mask) : probe_sum<ND>(S, vc, vd); };
    auto flush_final = [&](const ProbeShape<ND>& S, const bool final) {
        const float score = sum_of(S, f_vc, f_vd, f_mask);
        const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)f_doc;
        f_stage = 0;
        PS_COUNT(12)
        push_keys(key, f_alive && key > *thr && key < cs.upper, final);
    };
    // One call per tile: the flush in flight advances by one stage, or a new flush starts.  Whatever happens, exactly ONE gather goes out, at
    // one place, straight into f_r (idle: entry 0 of operand 0, a line every wave keeps hitting) — its register is then nowhere a temporary of
    // some other path, which the compiler could only protect with a wait for everything in flight.
    auto flush_service = [&](const bool final) {
        const unsigned long long* sptr = reinterpret_cast<const unsigned long long*>(sh + kShScores);
        unsigned long long gp = sptr[0];
        uint32_t gidx = 0u;
        if (f_stage) {  // uniform: one more operand's value is known
            const ProbeShape<ND> S = probe_shape<ND>(sh);
            const float thr_f = __uint_as_float(unorder_f32((uint32_t)(*thr >> 32)));  // NaN while there is no threshold: nothing is dropped
            bool last = true;
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i)
                if (f_stage == i + 1u) {  // uniform
                    f_vd[i] = posting_value(S.ts[i], f_r);
                    if (i + 1u < ND) {  // hits that can no longer reach the threshold drop out, the next operand's gather goes out
                        constexpr uint32_t zero = 0;
                        f_alive = f_alive && !(sum_of(S, f_vc, f_vd, f_mask) < thr_f);
                        if (stat && lane == 0) *stat += 2u * (uint32_t)__popcll(wballot(f_alive));
                        gp = sptr[i + 1u < ND ? i + 1u : zero];
                        gidx = (f_alive && (!OR || ((f_mask >> (i + 1u < ND ? i + 1u : zero)) & 1u))) ? f_idx[i + 1u < ND ? i + 1u : zero] : 0u;
                        last = false;
                    }
                }
            if (last) flush_final(S, final);
            else ++f_stage;
        } else if (rn >= 64u || (final && rn)) {  // uniform: start a flush — the first operand's gather goes out
            const uint32_t count = rn < 64u ? rn : 64u;
            probe_lds_fence();
            f_alive = lane < count;
            uint32_t raw = 0;
            const uint32_t slot = (rhead + lane) & (kPR - 1u);
            if (f_alive) {
                f_doc = rq[slot];
                raw = rq[kPR + slot];
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) f_idx[i] = rq[(2u + i) * kPR + slot];
            }
            if (OR) {
                f_mask = 0u;
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) f_mask |= f_idx[i] != 0xFFFFFFFFu ? 1u << i : 0u;
            }
            f_vc = posting_value(__uint_as_float(sh[kShCts]), (uint16_t)raw);
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i) f_vd[i] = __uint_as_float(sh[kShVmax + i]);
            if (stat && lane == 0) *stat += 2u * count;  // gathered bytes of the span
            gidx = (f_alive && (!OR || (f_mask & 1u))) ? f_idx[0] : 0u;
            f_stage = 1u;
            rhead = (rhead + count) & (kPR - 1u);
            rn -= count;
        }
        f_r = as_global(reinterpret_cast<const uint16_t*>((uintptr_t)gp))[gidx];
    };
    // The ranked queue is full while a flush is still in flight (warm-up, or a dense stretch of hits): 64 entries are scored on the spot —
    // all operands gathered at once and waited for.  (Its own code and its own registers: the pipelined flush has ONE place that issues a
    // gather, so that gather's register is never a temporary somewhere else that the compiler would have to wait on.)
    auto flush_sync = [&]() {
        const ProbeShape<ND> S = probe_shape<ND>(sh);
        probe_lds_fence();
        const uint32_t slot = (rhead + lane) & (kPR - 1u);
        const uint32_t doc = rq[slot];
        const float vc = posting_value(S.cts, (uint16_t)rq[kPR + slot]);
        float vd[ND];
        uint32_t mask = 0;
#pragma unroll
        for (uint32_t i = 0; i < ND; ++i) {
            const uint16_t* sp = reinterpret_cast<const uint16_t*>((uintptr_t) reinterpret_cast<const unsigned long long*>(sh + kShScores)[i]);
            const uint32_t ix = rq[(2u + i) * kPR + slot];
            if (ix != 0xFFFFFFFFu) mask |= 1u << i;
            vd[i] = posting_value(S.ts[i], as_global(sp)[(OR && ix == 0xFFFFFFFFu) ? 0u : ix]);
        }
        if (stat && lane == 0) *stat += 2u * ND * 64u;
        const float score = sum_of(S, vc, vd, mask);
        const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
        rhead = (rhead + 64u) & (kPR - 1u);
        rn -= 64u;
        push_keys(key, key > *thr && key < cs.upper, false);
    };
    // index of the doc at in-tile offset `rel` in bitmap operand i: rank directory entry of its 512-doc group + set bits of the group below it
    auto bitmap_rank = [&](const uint32_t i, const uint32_t rel) {
        const uint32_t g = rel >> kRankShift, wi = (rel >> 5) & 15u, below = (1u << (rel & 31u)) - 1u;
        const int full = (int)((1u << wi) - 1u);  // bit j: word j of the group lies entirely below the doc
        const uint32_t* tl = tile + i * kPTW;
        uint32_t acc = rank[i * kPRk + g] + (uint32_t)__popc(tl[rel >> 5] & below);
        const u32x4* gw = reinterpret_cast<const u32x4*>(tl + g * 16u);
#pragma unroll
        for (uint32_t v4 = 0; v4 < 4; ++v4) {
            const u32x4 x = gw[v4];
            acc += (uint32_t)__popc(x.x & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 0u, 1u));
            acc += (uint32_t)__popc(x.y & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 1u, 1u));
            acc += (uint32_t)__popc(x.z & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 2u, 1u));
            acc += (uint32_t)__popc(x.w & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 3u, 1u));
        }
        return acc;
    };
    // Take the LAST `cnt` (<= 64) entries of the unranked queue (their order is of no consequence: nothing has to move) — postings of the CURRENT tile, whose words / arrays are in LDS — into the
    // ranked queue.  Without array operands the entries are live hits already and only need their indices in the bitmap operands.  With array
    // operands they are postings that passed the bitmap operands: each is looked up in every array (binary search over the tile's sorted
    // offsets; the pads are the largest values), the ones found are the query's hits, and the strong ones among them are ranked.
    auto rank_some = [&](const uint32_t cnt) {
        while (rn + cnt > kPR) flush_sync();  // uniform, warm-up only
        probe_lds_fence();
        const uint32_t* const uqe = uq + (un - cnt);
        un -= cnt;
        if (NA == 0u && ND == 2u && cnt <= 32u) {  // uniform: the usual case — both operands at once, lanes 0-31 rank in operand 0, lanes 32-63 in operand 1
            const uint32_t el = lane & 31u, role = lane >> 5;
            if (el < cnt) {
                const uint32_t e = uqe[el];
                const uint32_t rel = e >> 16;  // doc - tile_lo
                const uint32_t slot = (rhead + rn + el) & (kPR - 1u);
                if (role == 0u) {
                    rq[slot] = tile_lo + rel;
                    rq[kPR + slot] = e & 0xFFFFu;
                }
                const bool here = !OR || ((tile[role * kPTW + (rel >> 5)] >> (rel & 31u)) & 1u) != 0u;  // (an OR's doc need not be in every operand)
                rq[(2u + role) * kPR + slot] = here ? bitmap_rank(role, rel) : 0xFFFFFFFFu;
            }
            rn += cnt;
        } else if (NA == 0u) {
            if (lane < cnt) {
                const uint32_t e = uqe[lane];
                const uint32_t rel = e >> 16;  // doc - tile_lo
                const uint32_t slot = (rhead + rn + lane) & (kPR - 1u);
                rq[slot] = tile_lo + rel;
                rq[kPR + slot] = e & 0xFFFFu;
#pragma unroll
                for (uint32_t i = 0; i < NB; ++i) {
                    const bool here = !OR || ((tile[i * kPTW + (rel >> 5)] >> (rel & 31u)) & 1u) != 0u;
                    rq[(2u + i) * kPR + slot] = here ? bitmap_rank(i, rel) : 0xFFFFFFFFu;
                }
            }
            rn += cnt;
        } else {
            uint32_t e = 0xFFFFFFFFu, pos[NA1];
            bool found = lane < cnt;
            if (found) e = uqe[lane];
            const uint32_t rel = e >> 16;
#pragma unroll
            for (uint32_t a = 0; a < NA; ++a) {
                const uint16_t* A = reinterpret_cast<const uint16_t*>(tile + (NB + a) * kPTW);
                uint32_t lo = 0u, len = a_cnt[a];  // uniform length: every lane takes the same number of steps
#ifdef VQ_PROBE_NO_SEARCH  // diagnostic build (wrong results): what the kernel takes without the lookups' dependent reads
                lo = rel & 7u;
                len = 0u;
#endif
                while (len > 1u) {
                    const uint32_t half = len >> 1;
                    lo = (uint32_t)A[lo + half] <= rel ? lo + half : lo;  // the last entry <= rel
                    len -= half;
                }
#ifdef VQ_PROBE_NO_SEARCH
                found = found && a_cnt[a] != 0u && ((uint32_t)A[lo] ^ rel) < 0x2000u;  // (one read; about a quarter of the candidates "found", as in the real query)
#else
                found = found && a_cnt[a] != 0u && (uint32_t)A[lo] == rel;
#endif
                pos[a] = lo;
            }
            const unsigned long long fm = wballot(found);
            hits += (unsigned long long)__popcll(fm);
            const bool live = found && (e & 0xFFFFu) >= raw_min;
            const unsigned long long lm = wballot(live);
            if (lm) {  // uniform
                if (live) {
                    const uint32_t slot = (rhead + rn + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))) & (kPR - 1u);
                    rq[slot] = tile_lo + rel;
                    rq[kPR + slot] = e & 0xFFFFu;
#pragma unroll
                    for (uint32_t i = 0; i < NB; ++i) rq[(2u + i) * kPR + slot] = bitmap_rank(i, rel);
#pragma unroll
                    for (uint32_t a = 0; a < NA; ++a) rq[(2u + NB + a) * kPR + slot] = (a_g0[a] * 8u + pos[a]) * 2u;  // (u16 index of the packed word's low half)
                }
                rn += (uint32_t)__popcll(lm);
            }
        }
    };
    // one round of 256 cover postings (lane l: four consecutive ones) against the tile in LDS
    uint32_t lo_rel = 0, width = 0;
    struct ProbeWords {
        uint32_t w[OR ? ND : 1u][4];  // AND: the operands' words ANDed; OR: every operand's word
    };
    auto probe_read = [&](const u32x4 e4) {  // the bitmap operands' words at the four postings of a lane
        const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
        ProbeWords pw;
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t a = (ee[c] >> 21) & (kPTW - 1u);
            if (OR) {
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) pw.w[OR ? i : 0u][c] = tile[i * kPTW + a];
            } else {
                pw.w[0][c] = 0xFFFFFFFFu;
                if (NB) {
                    pw.w[0][c] = tile[a];
#pragma unroll
                    for (uint32_t i = 1; i < NB; ++i) pw.w[0][c] &= tile[i * kPTW + a];
                }
            }
        }
        return pw;
    };
    auto probe_eval = [&](const u32x4 e4, const ProbeWords& pw) {
        const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
        unsigned long long lm[4];
        bool live[4];
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t rel = ee[c] >> 16;
            const bool in = (rel - lo_rel) < width;  // (the span's first and last tile are cut; a pad's offset 65535 is outside every tile)
            if (OR) {
                // the operands that hold the doc decide its bound; a cover posting in none of them is a hit the words' popcount has not seen
                uint32_t mask = 0;
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) mask |= ((pw.w[OR ? i : 0u][c] >> (rel & 31u)) & 1u) << i;
                uint32_t need = rmin[0];
#pragma unroll
                for (uint32_t m = 1; m < kMasks; ++m) need = mask == m ? rmin[OR ? m : 0u] : need;
                hits += (unsigned long long)__popcll(wballot(in && mask == 0u));
                live[c] = in && (ee[c] & 0xFFFFu) >= need;
                lm[c] = wballot(live[c]);
                continue;
            }
            const bool bit = ((pw.w[0][c] >> (rel & 31u)) & 1u) != 0u;
            // (ballots of the plain compares, joined as masks: a ballot of a joined bool costs two more vector instructions)
            const unsigned long long sm = NB ? (wballot(in) & wballot(bit)) : wballot(in);
            if (NA == 0u) {
                const bool strong = (ee[c] & 0xFFFFu) >= raw_min;
                hits += (unsigned long long)__popcll(sm);
                lm[c] = sm & wballot(strong);
                live[c] = in && bit && strong;
            } else {  // (the arrays decide what a hit is: rank_some counts them)
                lm[c] = sm;
                live[c] = in && bit;
            }
        }
        if (lm[0] | lm[1] | lm[2] | lm[3]) {  // uniform
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                if (live[c]) uq[un + (uint32_t)__popcll(lm[c] & ((1ull << lane) - 1ull))] = ee[c];  // (in-tile offset << 16 | raw score: the packed word itself)
                un += (uint32_t)__popcll(lm[c]);
            }
        }
    };

    uint32_t t = t_first;
    tile_lo = bitmap_base + (t_first << kProbeTileShift);
    auto next_arrays = [&](const uint32_t tt) {  // uniform: array operands' slices of tile tt out of their directory registers
#pragma unroll
        for (uint32_t a = 0; a < NA; ++a) {
            na_g0[a] = dir_at(adv[a], tt);
            na_nv[a] = dir_at(adv[a], tt + 1u) - na_g0[a];
        }
    };
    if (t < t_end) {
        next_arrays(t);
        issue_tile(t, dir_at(dirv, t), dir_at(dirv, t + 1u));
    }
    PS_AT(0)
    while (t < t_end) {  // uniform
        PS_COUNT(8)
        // ---- top of the tile: everything in flight lands here — one wait for all of it (vmcnt(0); the compiler's own bookkeeping sees the
        // instruction and knows of nothing in flight behind it, so none of ITS waits can fall behind the next tile's loads)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_setprio(3);  // a wave whose data has landed goes first until its next loads are out (nothing it does in between should queue behind other waves' arithmetic)
        const uint32_t rounds = pf_rounds, v0 = pf_v0, v1 = pf_v1;
        if (OR || rounds) {  // uniform
            if (OR) {  // every doc of an operand is a hit: the set bits of the operands' words, ORed (the cover's postings outside them are counted as they pass)
#pragma unroll
                for (uint32_t h = 0; h < kPNV; ++h) {
                    u32x4 u = wk[0][h];
#pragma unroll
                    for (uint32_t i = 1; i < NB; ++i) u |= wk[i][h];
                    ucnt += (uint32_t)__popc(u.x) + (uint32_t)__popc(u.y) + (uint32_t)__popc(u.z) + (uint32_t)__popc(u.w);
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) {
#pragma unroll
                for (uint32_t h = 0; h < kPNV; ++h) reinterpret_cast<u32x4*>(tile + i * kPTW)[h * 64u + lane] = wk[i][h];
                rank[i * kPRk + lane] = rk[i];
            }
#pragma unroll
            for (uint32_t a = 0; a < NA; ++a) {
#pragma unroll
                for (uint32_t h = 0; h < kPNV; ++h)
                    if (h * 64u < na_nv[a]) reinterpret_cast<u32x4*>(tile + (NB + a) * kPTW)[h * 64u + lane] = wk[NB + a][h];  // uniform: only the vectors that hold entries of the tile
                a_g0[a] = na_g0[a];
                a_cnt[a] = na_nv[a] * 8u;
            }
#pragma unroll
            for (uint32_t r = 0; r < kPMaxR; ++r) {
                cid[r] = nid[r];
                // (a use the compiler cannot rename away: it has to wait for the postings HERE, where everything in flight has landed — not
                //  later, behind the next tile's loads, where its only safe wait is for all of them)
                asm volatile("" : "+v"(cid[r]));
            }
        }
        // ---- first everything that CONSUMES a load of the last period (the flush in flight, the shared threshold word) ...
        {  // (every register a load of the last period wrote is touched here, where all of them have landed: behind this point the compiler
           //  knows of nothing in flight, and none of its waits can fall behind the next tile's loads)
            uint32_t fr = f_r, glo = (uint32_t)g_prev, ghi = (uint32_t)(g_prev >> 32);
            asm volatile("" : "+v"(fr), "+v"(glo), "+v"(ghi), "+v"(dirv));
#pragma unroll
            for (uint32_t a = 0; a < NA; ++a) asm volatile("" : "+v"(adv[a]));
            f_r = (uint16_t)fr;
            g_prev = ((unsigned long long)ghi << 32) | glo;
        }
        PS_AT(1)
        if (lane == 0 && g_prev > *thr) *thr = g_prev;  // what other spans of the query have published (QHeader::gthr), asked for a tile ago
        const bool more = t + 1u < t_end;
        uint32_t ng0 = 0, ng1 = 0;  // the next tile's slice of the cover (directory entries)
        if (more) {  // uniform
            if (t + 2u - dir_base >= 64u) {  // the directory slices are used up (62 tiles): the next ones (a wait, once per 62 tiles)
                dir_base = t + 1u;
                dirv = load_dir(cgdir);
#pragma unroll
                for (uint32_t a = 0; a < NA; ++a) adv[a] = load_dir(d_agd[a]);
            }
            ng0 = dir_at(dirv, t + 1u);
            ng1 = dir_at(dirv, t + 2u);
            next_arrays(t + 1u);
        }
        if (f_stage || rn >= 64u) PS_COUNT(11)
        flush_service(false);  // takes the gather issued a tile ago, issues the next one (the LAST consumer of an old load: behind it only new ones go out)
        PS_AT(4)
        // ---- ... then everything the NEXT tile needs is asked for, and nothing below waits for any of it
        if (more) issue_tile(t + 1u, ng0, ng1);  // uniform
        else pf_rounds = 0;
        if (lane == 0) g_prev = __hip_atomic_load(gthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_setprio(0);
        probe_lds_fence();
        {
            const unsigned long long tn = *thr;
            if ((uint32_t)(tn >> 32) != (uint32_t)(thr_seen >> 32)) take_raw_min(__uint_as_float(unorder_f32((uint32_t)(tn >> 32))));  // uniform
            thr_seen = tn;
            ++tiles_since_merge;
        }
        PS_AT(5)
#ifdef VQ_PROBE_STREAM_ONLY  // diagnostic build: the loads and the LDS fill only (what the memory side alone takes)
        hits += (unsigned long long)__popcll(wballot((cid[0].x ^ cid[1].y) == 0x12345u));  // (keeps the cover loads alive)
        if (rounds && span == 0xFFFFFFFFu) {
#else
        if (rounds) {  // uniform
#endif
            // ---- the cover's postings of the tile against the operands' words
            const uint32_t tile_end = tile_lo + kPT;
            const uint32_t tile_hi = (tile_end > tile_lo && tile_end < span_hi) ? tile_end : span_hi;
            const uint32_t lo_bound = tile_lo > span_lo ? tile_lo : span_lo;
            lo_rel = lo_bound - tile_lo;
            width = tile_hi - lo_bound;
            {  // the register rounds: every LDS read of the tile's postings goes out first, then the few hits are picked up
                ProbeWords pw[kPMaxR];
#pragma unroll
                for (uint32_t r = 0; r < kPMaxR; ++r)
                    if (r < rounds) pw[r] = probe_read(cid[r]);  // uniform
#pragma unroll
                for (uint32_t r = 0; r < kPMaxR; ++r)
                    if (r < rounds) {  // uniform
                        PS_COUNT(9)
                        probe_eval(cid[r], pw[r]);
                    }
            }
            for (uint32_t r = kPMaxR; r < rounds; ++r) {  // a dense stretch of the cover: further rounds are fetched on the spot
                while (un >= kPU - 256u) rank_some(64u);  // uniform: room for another round
                const uint32_t v = v0 + r * 64u + lane;
                u32x4 e4 = kSent;
                if (v < v1) e4 = cc4[v];
                PS_COUNT(9)
                const ProbeWords pw = probe_read(e4);
                probe_eval(e4, pw);
            }
            PS_AT(2)
            while (un) {  // uniform: the tile's queued postings are looked up / ranked while its words are still in LDS
                PS_COUNT(10)
                rank_some(un < 64u ? un : 64u);
            }
            PS_AT(3)
        }
        ++t;
        tile_lo += kPT;
    }
    while (f_stage || rn) flush_service(true);  // uniform: the flush pipeline drains (these gathers are waited for where they are used)
    __syncthreads();
    cand_prune(cs, top_k);
    {
        const uint32_t cn = *cand_n;
        unsigned long long* out = span_keys + (size_t)keys_base + (size_t)span * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cand[i] : 0ull;
    }
    if (OR) {
        uint32_t total;
        (void)wave_excl_scan_u32(ucnt, &total);
        hits += (unsigned long long)total;
    }
    if (lane == 0 && hits) atomicAdd(&num_hits[q], hits);
    if (lane == 0 && H->stat_off && lds[4]) atomicAdd(&num_hits[H->stat_off], (unsigned long long)lds[4]);
    PS_AT(6)
    PS_FLUSH
}

// which (query, span) a workgroup of the launch runs
__device__ __forceinline__ void probe_item(const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq, uint32_t* q, uint32_t* span) {
    uint32_t lo = 0, hi = nq;
    const uint32_t wg = blockIdx.x;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (span_base[mid] <= wg) lo = mid;
        else hi = mid;
    }
    *span = blockIdx.x - span_base[lo];
    *q = qmap[lo];
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_scan_probe(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                                                               const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq,
                                                                                               uint32_t cand_cap, unsigned long long* __restrict__ span_keys,
                                                                                               unsigned long long* __restrict__ num_hits) {
    uint32_t q, span;
    probe_item(span_base, qmap, nq, &q, &span);
    const uint8_t* blob = blobs + blob_off[q];
    const uint32_t n = as_const<QHeader>(blob)->simple_n;
    if ((as_const<QHeader>(blob)->simple_flags >> 27) & 1u) return;  // an OR: k_scan_probe_or's (same grid)
    const uint32_t na = (uint32_t)__popc((as_const<QHeader>(blob)->simple_flags >> 12) & 0xFu);  // operands probed as 16-bit arrays
    if (n == 2u) {
        if (na == 0u) probe_body<1, 0>(blob, span, q, cand_cap, span_keys, num_hits);
        else probe_body<1, 1>(blob, span, q, cand_cap, span_keys, num_hits);
    } else if (n == 3u) {
        if (na == 0u) probe_body<2, 0>(blob, span, q, cand_cap, span_keys, num_hits);
        else if (na == 1u) probe_body<2, 1>(blob, span, q, cand_cap, span_keys, num_hits);
        else probe_body<2, 2>(blob, span, q, cand_cap, span_keys, num_hits);
    } else {
        if (na == 0u) probe_body<3, 0>(blob, span, q, cand_cap, span_keys, num_hits);
        else if (na == 1u) probe_body<3, 1>(blob, span, q, cand_cap, span_keys, num_hits);
        else if (na == 2u) probe_body<3, 2>(blob, span, q, cand_cap, span_keys, num_hits);
        else probe_body<3, 3>(blob, span, q, cand_cap, span_keys, num_hits);
    }
}

// The OR form of the same scan (simple_flags bit 27): an OR of 2 or 3 leaves with a term slot each whose sparsest operand streams as the cover
// and whose other operands are bitmap words.  num_hits is the union: the set bits of the operands' ORed words plus the cover's postings
// outside them.  Scored are the docs that hold the COVER — with whatever operands hold them too, each such set with its own bound.  Docs
// WITHOUT the cover are counted, never scored: exact as long as none of them can reach the request's k-th best score — their scores are
// bounded by the OR formula on the operands' list maxima (CompiledQuery::or_skip_bound), and the host checks the k-th key of the finished
// request against that bound (finish_batch), running the request again on k_scan_simple when the check fails (an OR whose best hits lack
// its rarest term).  Its own kernel: its register needs are not the AND's.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_scan_probe_or(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off,
                                                                                                  const uint32_t* __restrict__ span_base, const uint32_t* __restrict__ qmap, uint32_t nq,
                                                                                                  uint32_t cand_cap, unsigned long long* __restrict__ span_keys,
                                                                                                  unsigned long long* __restrict__ num_hits) {
    uint32_t q, span;
    probe_item(span_base, qmap, nq, &q, &span);
    const uint8_t* blob = blobs + blob_off[q];
    const uint32_t n = as_const<QHeader>(blob)->simple_n;
    if (!((as_const<QHeader>(blob)->simple_flags >> 27) & 1u)) return;
    if (n == 2u) probe_body<1, 0, true>(blob, span, q, cand_cap, span_keys, num_hits);
    else probe_body<2, 0, true>(blob, span, q, cand_cap, span_keys, num_hits);  // (compile.cpp sends ORs of 2 or 3 leaves: a fourth operand's eight bounds spill)
}

// max_nd: most dense operands of a query of the launch (sizes the LDS tile area); any_and / any_or: which kinds of query the launch holds
void launch_scan_probe(hipStream_t st, uint32_t max_nd, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base, const uint32_t* qmap,
                       uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, bool any_and, bool any_or) {
    if (!total_spans) return;
    if (any_and)
        hipLaunchKernelGGL(k_scan_probe, dim3(total_spans), dim3(64), scan_probe_lds_bytes(cand_cap, max_nd), st, blobs, blob_off, span_base, qmap, nq, cand_cap, span_keys, num_hits);
    if (any_or)
        hipLaunchKernelGGL(k_scan_probe_or, dim3(total_spans), dim3(64), scan_probe_lds_bytes(cand_cap, max_nd), st, blobs, blob_off, span_base, qmap, nq, cand_cap, span_keys, num_hits);
}

}  // namespace vq
