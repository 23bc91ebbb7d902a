// k_scan_ring — the AND-probe scan (scan_probe.hip: cover postings tested against the dense operands' bitmap words) as a PERSISTENT kernel
// whose memory stream is taken out of the computing waves: one workgroup per CU = kRingL LOADER waves + C CONSUMER waves around rings of
// tile slots in LDS.
//
//   consumer  owns (query, span) items drawn from a work counter.  It walks its span's tiles (32768 docs), asks for the next non-empty ones —
//             a REQUEST is eight global pointers (the operands' words and rank entries of the tile, the cover's ids and scores) plus the
//             length of the tile's slice of the cover — up to S tiles ahead, and works on a tile once a loader has published it: the cover's
//             postings of the tile test their bits in the operands' words, live hits are ranked, queued and scored exactly as in
//             k_scan_probe (same bound, same flush pipeline, same shared top-k pool: bit-identical results).  It never waits for a load of
//             the stream, and looks at its own few loads (score gathers, the query's threshold word) only every kRingSvc-th tile.
//   loader    serves its consumers' request queues round-robin: 5 * MAXND + 3 LDS-DMA pieces per tile (global_load_lds_dwordx4 / _dword:
//             no VGPR destination, no ds_write), ONE asm statement, published to the consumer behind a counted s_waitcnt vmcnt that
//             leaves kRingM younger tiles in flight.  The compiler's own wait bookkeeping never sees these loads and never drains them.
//             (tools/glds_ring.hip measured the structure alone: one loader wave per CU streams 6.4 TB/s into the rings.)
//
// Hand-offs are LDS words between waves of one workgroup (in-order LDS per wave, counts only ever grow): req_count[c] (consumer ->
// loader: requests posted, which also frees the slot the request names), full_count[c] (loader -> consumer: tiles landed), done[c].
// Every wait on another wave is bounded (kRingSpin polls): a wave that gives up marks the launch failed (err word, poisoned hit count)
// and leaves — the grid always drains.
//
// What the first version taught (profiles/r04_ring_*): the scan is bound by INSTRUCTIONS as much as by bytes — 4.3 G wave-instructions per
// 1024 queries against k_scan_probe's 2.1 G, with 166 spilled SGPRs reloaded inside the tile loop.  Hence: the tile loop is a function of
// its own (ring_tiles, not inlined: its registers are its own), what a span needs only at its start and end lives in an LDS context block,
// and everything rare (candidate buffer, pool merge, raw_min, the synchronous flush) sits behind calls.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "device_types.hpp"
#include "kernel_common.hpp"
#include "kernels.hpp"
#include "probe_common.hpp"

#ifndef VQ_RING_L
#define VQ_RING_L 2
#endif
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // LDS byte addresses (32 bits) become address_space(3) pointers (32 bits)

namespace vq {

constexpr uint32_t kGT = 1u << kProbeTileShift;        // docs per tile (32768)
constexpr uint32_t kGTW = kGT / 32;                     // bitmap words per tile and dense operand (1024)
constexpr uint32_t kGRk = kGT >> kRankShift;            // rank directory entries per tile and dense operand (64)
constexpr uint32_t kGDir = kProbeTileShift - kTileDirShift;  // log2 of the cover's directory entries per tile
constexpr uint32_t kRingL = VQ_RING_L;   // loader waves per workgroup (loader l serves the consumers c with c % kRingL == l)
constexpr uint32_t kRingM = 3;           // tiles a loader keeps in flight behind the one it is issuing
constexpr uint32_t kRingMaxS = 3;        // slots per consumer, at most
constexpr uint32_t kRingRounds = 2;      // rounds of 256 cover postings a slot holds (a denser tile fetches the rest itself)
constexpr uint32_t kRU = 64 + 256;       // unranked queue: live hits of the current tile, (doc - tile_lo) << 16 | raw f16 score of the cover
constexpr uint32_t kRR = 64;             // ranked queue (a ring)
constexpr uint32_t kRCand = 64;          // candidate buffer: one key per lane (top_k <= kPoolMaxK = 32)
constexpr uint32_t kReqWords = 24;       // 8 pointers | pad, pad, pad, nv | pad
constexpr uint32_t kRingSpin = 1u << 22;
constexpr uint32_t kRingSvc = 4;         // a consumer looks at its own loads in flight (flush gathers, the query's threshold word) every 4th tile
// LDS map (u32): ctl[64] | consumer 0 | consumer 1 | ...;  ctl: req_count @0, full_count @16, done @32
// consumer: reqs[kRingMaxS][kReqWords] | ctx[64] | misc[8] | shape[32] | uq[kRU] | rq: rdoc[kRR] rraw[kRR] ridx[MAXND][kRR] | cand[2 * kRCand] | pad | slots[S][slot]
// slot: words[MAXND][kGTW] | rank[MAXND][kGRk] | ids[256 * rounds] | scores[128 * rounds]
constexpr uint32_t kRCtl = 64;
constexpr uint32_t kOCtx = kRingMaxS * kReqWords, kOMisc = kOCtx + 64, kOShape = kOMisc + 8, kOU = kOShape + 32, kOR = kOU + kRU;
__host__ __device__ constexpr uint32_t ring_off_cand(uint32_t maxnd) { return kOR + (2 + maxnd) * kRR; }
__host__ __device__ constexpr uint32_t ring_priv_words(uint32_t maxnd) { return (ring_off_cand(maxnd) + 2 * kRCand + 63u) & ~63u; }
__host__ __device__ constexpr uint32_t ring_slot_words(uint32_t maxnd) { return maxnd * (kGTW + kGRk) + kRingRounds * 256 + kRingRounds * 128; }
__host__ __device__ constexpr uint32_t ring_cons_words(uint32_t maxnd, uint32_t S) { return ring_priv_words(maxnd) + S * ring_slot_words(maxnd); }
// ctx words (what a span needs at its start, at its end and on its rare paths; the tile loop keeps none of it in registers)
constexpr uint32_t kCGthr = 2, kCPool = 4, kCDocs = 6, kCScores = 8, kCDir = 10, kCLen = 12, kCTFirst = 13, kCTEnd = 14, kCBase = 15, kCTopK = 16, kCStat = 17,
                   kCUpper = 18, kCQ = 20, kCSpan = 21, kCKeys = 22, kCFailed = 23, kCNReq = 26, kCNDone = 27, kCSReq = 28, kCSDone = 29, kCRqBase = 32, kCRqMul = 48;
// misc words: thr (u64) @0, cand_n @2, stat @4, raw_min @5, thr_seen (score half) @6

uint32_t scan_ring_slots(uint32_t maxnd, uint32_t C) {  // slots per consumer that fit 160 KiB
    const uint32_t avail = (160u * 1024u / 4u - kRCtl) / C;
    if (avail <= ring_priv_words(maxnd)) return 0;
    const uint32_t s = (avail - ring_priv_words(maxnd)) / ring_slot_words(maxnd);
    return s < kRingMaxS ? s : kRingMaxS;
}
size_t scan_ring_lds_bytes(uint32_t maxnd, uint32_t C, uint32_t S) { return (size_t)(kRCtl + C * ring_cons_words(maxnd, S)) * 4; }

// ---- LDS through the LDS address space, by byte address: inside a function that is not inlined a generic pointer would be a flat access
// (and a volatile one a flat_load sc0 sc1 + s_waitcnt vmcnt(0) — it would drain every DMA in flight)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t L32;
typedef __attribute__((address_space(3))) unsigned long long L64;
typedef __attribute__((address_space(3))) u32x2 L2x32;
typedef __attribute__((address_space(3))) u32x4 L4x32;
__device__ __forceinline__ L32* L(uint32_t byte) { return (L32*)byte; }
__device__ __forceinline__ L64* LL(uint32_t byte) { return (L64*)byte; }
__device__ __forceinline__ uint32_t lds_ld(uint32_t byte) { return *(const volatile L32*)byte; }
__device__ __forceinline__ void lds_st(uint32_t byte, uint32_t v) { *(volatile L32*)byte = v; }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ unsigned long long lds_ptr(uint32_t byte) { return *(const L64*)byte; }

#ifdef VQ_RING_NT
#define VQ_RING_POLICY " nt"
#else
#define VQ_RING_POLICY ""
#endif
template <int N>
__device__ __forceinline__ void vmcnt_imm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <uint32_t K>
__device__ __forceinline__ void wait_all_but_tiles(uint32_t m) {  // every DMA but those of the m (<= kRingM) youngest tiles has landed
    static_assert(kRingM == 3 && 3 * K <= 63, "cases");
    if (m >= 3u) vmcnt_imm<3 * K>();
    else if (m == 2u) vmcnt_imm<2 * K>();
    else if (m == 1u) vmcnt_imm<K>();
    else vmcnt_imm<0>();
}

// One tile's pieces, ONE statement (M0 = wave-uniform LDS byte address of the destination, saved and restored once; destination = M0 +
// instruction offset + lane * size, source = SGPR base + VGPR offset + instruction offset: tools/glds_ring.hip part A):
//   words of operand i: 4 x 1 KiB to slot + i * 4096; rank entries of operand i: 256 B to slot + MAXND * 4096 + i * 256;
//   cover ids: 2 x 1 KiB, cover scores: 1 KiB (sources clamped per lane: vid0, vid1, vsc)
#define VQ_G16(v, p, off) "global_load_lds_dwordx4 " v ", " p " offset:" off VQ_RING_POLICY "\n\t"
#define VQ_G4(v, p) "global_load_lds_dword " v ", " p VQ_RING_POLICY "\n\t"
#define VQ_M0(slot, off) "s_add_u32 m0, " slot ", " off "\n\ts_nop 0\n\t"
#define VQ_W4(v, p) VQ_G16(v, p, "0") VQ_G16(v, p, "1024") VQ_G16(v, p, "2048") VQ_G16(v, p, "3072")
template <uint32_t MAXND>
__device__ __forceinline__ void dma_tile(uint32_t slot, unsigned long long pw0, unsigned long long pw1, unsigned long long pw2, unsigned long long pr0, unsigned long long pr1,
                                         unsigned long long pr2, unsigned long long pi, unsigned long long ps, uint32_t v16, uint32_t v4, uint32_t vid0, uint32_t vid1, uint32_t vsc) {
    uint32_t keep;
    if (MAXND == 1u) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     VQ_M0("%1", "0") VQ_W4("%6", "%2")
                     VQ_M0("%1", "4096") VQ_G4("%7", "%3")
                     VQ_M0("%1", "4352") VQ_G16("%8", "%4", "0")
                     VQ_M0("%1", "5376") VQ_G16("%9", "%4", "0")
                     VQ_M0("%1", "6400") VQ_G16("%10", "%5", "0")
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(slot), "s"(pw0), "s"(pr0), "s"(pi), "s"(ps), "v"(v16), "v"(v4), "v"(vid0), "v"(vid1), "v"(vsc)
                     : "memory", "scc");
    } else if (MAXND == 2u) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     VQ_M0("%1", "0") VQ_W4("%8", "%2")
                     VQ_M0("%1", "4096") VQ_W4("%8", "%3")
                     VQ_M0("%1", "8192") VQ_G4("%9", "%4")
                     VQ_M0("%1", "8448") VQ_G4("%9", "%5")
                     VQ_M0("%1", "8704") VQ_G16("%10", "%6", "0")
                     VQ_M0("%1", "9728") VQ_G16("%11", "%6", "0")
                     VQ_M0("%1", "10752") VQ_G16("%12", "%7", "0")
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(slot), "s"(pw0), "s"(pw1), "s"(pr0), "s"(pr1), "s"(pi), "s"(ps), "v"(v16), "v"(v4), "v"(vid0), "v"(vid1), "v"(vsc)
                     : "memory", "scc");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     VQ_M0("%1", "0") VQ_W4("%10", "%2")
                     VQ_M0("%1", "4096") VQ_W4("%10", "%3")
                     VQ_M0("%1", "8192") VQ_W4("%10", "%4")
                     VQ_M0("%1", "12288") VQ_G4("%11", "%5")
                     VQ_M0("%1", "12544") VQ_G4("%11", "%6")
                     VQ_M0("%1", "12800") VQ_G4("%11", "%7")
                     VQ_M0("%1", "13056") VQ_G16("%12", "%8", "0")
                     VQ_M0("%1", "14080") VQ_G16("%13", "%8", "0")
                     VQ_M0("%1", "15104") VQ_G16("%14", "%9", "0")
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(slot), "s"(pw0), "s"(pw1), "s"(pw2), "s"(pr0), "s"(pr1), "s"(pr2), "s"(pi), "s"(ps), "v"(v16), "v"(v4), "v"(vid0), "v"(vid1), "v"(vsc)
                     : "memory", "scc");
    }
}
template <uint32_t MAXND>
__host__ __device__ constexpr uint32_t ring_pieces() { return 5u * MAXND + 3u; }  // LDS-DMA pieces per tile

// Diagnostic build only (make stamp): where the waves' time goes (s_memtime ticks summed over all waves) and event counts
#ifdef VQ_STAMP
__device__ unsigned long long g_ring_stamp[32];
#define RS_INIT                                             \
    unsigned long long _st0 = __builtin_amdgcn_s_memtime(); \
    unsigned long long _acc[32] = {0};
#define RS_AT(k)                                                \
    {                                                           \
        unsigned long long _st1 = __builtin_amdgcn_s_memtime(); \
        _acc[k] += _st1 - _st0;                                 \
        _st0 = _st1;                                            \
    }
#define RS_COUNT(k) _acc[k] += 1ull;
#define RS_FLUSH(lo, hi)                                                                                  \
    if ((threadIdx.x & 63u) == 0) {                                                                       \
        _Pragma("unroll") for (int _k = lo; _k < hi; ++_k) if (_acc[_k]) atomicAdd(&g_ring_stamp[_k], _acc[_k]); \
    }
void debug_read_ring_stamps(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_stamp), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_stamp), z, sizeof z);
    }
}
#else
#define RS_INIT
#define RS_AT(k)
#define RS_COUNT(k)
#define RS_FLUSH(lo, hi)
#endif

// ------------------------------------------------------------------------------------------------ loader
template <uint32_t MAXND, uint32_t C>
__device__ __forceinline__ void ring_loader(const uint32_t ctl, const uint32_t S, const uint32_t me, uint32_t* __restrict__ err) {
    constexpr uint32_t K = ring_pieces<MAXND>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cons_bytes = ring_cons_words(MAXND, S) * 4u;
    __builtin_amdgcn_s_setprio(3);
    RS_INIT
    uint32_t issued = 0, nslot = 0;          // lane c: tiles issued for consumer c, the slot its next tile goes to
    uint32_t fifo = 0, head = 0, npend = 0;  // issued, unpublished tiles: lane (head + i) & 7 holds consumer << 24 | its count after the tile
    uint32_t rr = 0, idle = 0;
    const uint32_t v16 = lane * 16u, v4 = lane * 4u;
    const bool mine = lane < C && lane % kRingL == me;
    while (true) {
        const uint32_t rc = lane < C ? lds_ld(ctl + lane * 4u) : 0u;
        const uint32_t dn = lane < C ? lds_ld(ctl + (32u + lane) * 4u) : 1u;
        probe_lds_fence();
        const unsigned long long want = wballot(mine && rc != issued);
        if (want) {  // uniform
            idle = 0;
            const unsigned long long hi = want & ~((1ull << rr) - 1ull);  // round-robin: the first wanting consumer at or behind rr
            const uint32_t c = (uint32_t)__builtin_ctzll(hi ? hi : want);
            rr = c + 1u;  // (a mask shifted by >= C lanes is empty: the search wraps)
            const uint32_t n = rdl(issued, c);
            const uint32_t s = rdl(nslot, c);
            const uint32_t cons = ctl + kRCtl * 4u + c * cons_bytes;
            const uint32_t rq = cons + s * (kReqWords * 4u);
            // the request: lane j < 8 reads pointer j (0-2 words, 3-5 rank entries, 6 ids, 7 scores), every lane the count of valid 16-byte
            // vectors of the cover's slice
            const uint32_t pj = lane & 7u;
            const u32x2 pp = *(const volatile L2x32*)(rq + 8u * pj);
            const uint32_t nv = lds_ld(rq + 19u * 4u);
            probe_lds_fence();
            const uint32_t nvm1 = uni(nv) - 1u;
            const uint32_t slot = cons + (ring_priv_words(MAXND) + s * ring_slot_words(MAXND)) * 4u;
            auto ptr = [&](uint32_t j) { return ((unsigned long long)rdl(pp.y, j) << 32) | rdl(pp.x, j); };
            // the cover's slice: lanes behind its last vector repeat that vector (no byte is fetched that the tile does not own)
            const uint32_t vid0 = (lane < nvm1 ? lane : nvm1) * 16u;
            const uint32_t l1 = lane + 64u;
            const uint32_t vid1 = (l1 < nvm1 ? l1 : nvm1) * 16u;
            const uint32_t nsm1 = nvm1 >> 1;  // last valid 16-byte vector of the scores (8 f16 each; the slice starts at a multiple of 8 postings)
            const uint32_t vsc = (lane < nsm1 ? lane : nsm1) * 16u;
            dma_tile<MAXND>(slot, ptr(0), MAXND >= 2u ? ptr(1) : 0ull, MAXND >= 3u ? ptr(2) : 0ull, ptr(3), MAXND >= 2u ? ptr(4) : 0ull, MAXND >= 3u ? ptr(5) : 0ull, ptr(6),
                            ptr(7), v16, v4, vid0, vid1, vsc);
            RS_COUNT(19)
            if (lane == c) {
                issued = n + 1u;
                nslot = s + 1u == S ? 0u : s + 1u;
            }
            if (npend == kRingM) {  // uniform: the oldest pending tile has kRingM younger ones behind it
                RS_AT(17)
                wait_all_but_tiles<K>(kRingM);
                RS_AT(18)
                const uint32_t e = rdl(fifo, head & 7u);
                if (lane == 0) lds_st(ctl + (16u + (e >> 24)) * 4u, e & 0xFFFFFFu);
                ++head;
                --npend;
            }
            if (lane == ((head + npend) & 7u)) fifo = (c << 24) | ((n + 1u) & 0xFFFFFFu);
            ++npend;
            RS_AT(17)
        } else if (npend) {  // nothing to issue: publish what is in flight, oldest first
            RS_AT(16)
            wait_all_but_tiles<K>(npend - 1u);
            const uint32_t e = rdl(fifo, head & 7u);
            if (lane == 0) lds_st(ctl + (16u + (e >> 24)) * 4u, e & 0xFFFFFFu);
            ++head;
            --npend;
            RS_AT(20)
        } else {
            if (!wballot(mine && !dn)) break;  // every consumer of this loader is done (a consumer posts nothing behind its done word)
            __builtin_amdgcn_s_sleep(2);
            RS_COUNT(21)
            if (++idle > 4u * kRingSpin) {
                if (lane == 0) atomicAdd(err, 1u);
                break;
            }
            RS_AT(16)
        }
    }
    vmcnt_imm<0>();
    RS_FLUSH(16, 24)
}

// ------------------------------------------------------------------------------------------------ consumer: the rare paths (not inlined)
// the span's candidate buffer (<= 64 keys, any order) sorted across the wave; the best k stay, the k-th becomes the threshold (and is
// published to / adopted from the query's threshold word)
__device__ __forceinline__ void ring_cand_prune(const CandState& cs, uint32_t k, const uint32_t lane) {
    probe_lds_fence();
    uint32_t n = *cs.n;
    n = n < kRCand ? n : kRCand;
    if (n <= k) {  // uniform
        if (lane == 0) *cs.n = n;
        probe_lds_fence();
        return;
    }
    unsigned long long key = lane < n ? cs.cand[lane] : 0ull;
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
    if (lane < k) cs.cand[lane] = key;
    unsigned long long t = shfl_u64(key, k - 1u);
    if (lane == 0) {
        *cs.n = k;
        if (cs.gthr) {
            const unsigned long long other = atomicMax(cs.gthr, t);
            t = other > t ? other : t;
        }
        *cs.thr = t;
    }
    probe_lds_fence();
}

// the consumer's candidate state out of its LDS block (rare paths only: generic pointers into LDS are flat accesses there)
template <uint32_t MAXND>
__device__ __forceinline__ CandState ring_cand_state(uint32_t* my) {
    CandState cs{reinterpret_cast<unsigned long long*>(my + ring_off_cand(MAXND)), my + kOMisc + 2, reinterpret_cast<unsigned long long*>(my + kOMisc), kRCand,
                 reinterpret_cast<unsigned long long*>((uintptr_t) * reinterpret_cast<unsigned long long*>(my + kOCtx + kCGthr))};
    cs.upper = *reinterpret_cast<unsigned long long*>(my + kOCtx + kCUpper);
    return cs;
}

// raw_min for the threshold the consumer holds now (kept in misc[5]; recomputed when the threshold's score has moved)
template <uint32_t ND>
__device__ __noinline__ uint32_t ring_refresh_raw_min(uint32_t* my) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* misc = my + kOMisc;
    const unsigned long long tn = *reinterpret_cast<unsigned long long*>(misc);
    if ((uint32_t)(tn >> 32) != misc[6]) {  // uniform
        const uint32_t rm = probe_raw_min<ND>(my + kOShape, __uint_as_float(unorder_f32((uint32_t)(tn >> 32))), lane);
        if (lane == 0) {
            misc[5] = rm;
            misc[6] = (uint32_t)(tn >> 32);
        }
        probe_lds_fence();
    }
    return misc[5];
}

// keys of scored hits -> the candidate buffer (pruned when it is full; merged into the query's pool when `merge`); -> raw_min
template <uint32_t ND, uint32_t MAXND>
__device__ __noinline__ uint32_t ring_push_keys(uint32_t* my, const unsigned long long key, bool pending, const bool merge) {
    const uint32_t lane = threadIdx.x & 63u;
    const CandState cs = ring_cand_state<MAXND>(my);
    const uint32_t top_k = my[kOCtx + kCTopK];
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
        probe_lds_fence();
        if (!wballot(pending)) break;
        ring_cand_prune(cs, top_k, lane);
    }
    uint8_t* const pool = reinterpret_cast<uint8_t*>((uintptr_t) * reinterpret_cast<unsigned long long*>(my + kOCtx + kCPool));
    if (pool && merge) {  // uniform
        ring_cand_prune(cs, top_k, lane);  // (the merge takes the buffer's first 32 keys: they must be its best)
        probe_pool_merge<false>(cs, top_k, pool, lane);
    }
    return ring_refresh_raw_min<ND>(my);
}

// The ranked queue is full while a flush is still in flight (warm-up, or a dense stretch of hits): its 64 entries at `rhead` are scored on
// the spot — all operands gathered at once and waited for.  -> raw_min
template <uint32_t ND, uint32_t MAXND>
__device__ __noinline__ uint32_t ring_flush_sync(uint32_t* my, const uint32_t rhead) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* sh = my + kOShape;
    uint32_t* rq = my + kOR;
    const ProbeShape<ND> Sh = probe_shape<ND>(sh);
    probe_lds_fence();
    const uint32_t slot = (rhead + lane) & (kRR - 1u);
    const uint32_t doc = rq[slot];
    const float vc = posting_value(Sh.cts, (uint16_t)rq[kRR + slot]);
    float vd[ND];
#pragma unroll
    for (uint32_t i = 0; i < ND; ++i) {
        const uint16_t* sp = reinterpret_cast<const uint16_t*>((uintptr_t) reinterpret_cast<const unsigned long long*>(sh + kShScores)[i]);
        vd[i] = posting_value(Sh.ts[i], as_global(sp)[rq[(2u + i) * kRR + slot]]);
    }
    if (my[kOCtx + kCStat] && lane == 0) my[kOMisc + 4] += 2u * ND * 64u;
    const float score = probe_sum<ND>(Sh, vc, vd);
    const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)doc;
    const unsigned long long thr = *reinterpret_cast<unsigned long long*>(my + kOMisc), upper = *reinterpret_cast<unsigned long long*>(my + kOCtx + kCUpper);
    return ring_push_keys<ND, MAXND>(my, key, key > thr && key < upper, false);
}

// ------------------------------------------------------------------------------------------------ consumer: a span's setup
// The query's shape (as in k_scan_probe) into the shape words, what the tile loop and the rare paths need into the context block.
template <uint32_t ND>
__device__ __noinline__ void ring_setup(const uint8_t* __restrict__ blob, const uint32_t span, const uint32_t q, uint32_t* my) {
    const uint32_t lane = threadIdx.x & 63u;
    const VQ_CONST QHeader* H = as_const<QHeader>(blob);
    const VQ_CONST DList* gl = as_const<DList>(blob + H->off_lists);
    const VQ_CONST DOp* gops = as_const<DOp>(blob + H->off_ops);
    const uint32_t sflags = H->simple_flags;
    constexpr uint32_t n = ND + 1u;
    uint32_t* const ctx = my + kOCtx;
    uint32_t* const misc = my + kOMisc;
    uint32_t* const sh = my + kOShape;
    unsigned long long* const gthr = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(blob) + offsetof(QHeader, gthr));
    const uint32_t* cdocs = nullptr;
    const uint16_t* cscores = nullptr;
    const uint32_t* ctdir = nullptr;
    uint32_t clen = 0;
    unsigned long long rq_base = 0ull;  // lane j < 8: pointer j of a request = rq_base + x * rq_mul, x = the tile (j < 6) or the slice's first vector (j >= 6)
    uint32_t rq_mul = 0u;
    {
        const uint32_t ck = (uint32_t)__ffs((int)((sflags >> 8) & 0xFu)) - 1u;
        uint32_t role_of[4] = {0u, 0u, 0u, 0u};  // leaf k -> 0 = cover, 1 + i = dense operand i
        uint32_t i = 0;
#pragma unroll
        for (uint32_t k = 0; k < n; ++k) {
            const VQ_CONST DList& d = gl[gops[k].list_begin];
            if (k == ck) {
                cdocs = d.docs;
                cscores = d.scores;
                ctdir = d.tile_dir;
                clen = d.len;
                if (lane == 0) {
                    sh[kShCts] = __float_as_uint(d.term_score);
                    sh[kShPrunable] = (d.term_score > 0.0f && d.max_raw < 0x7C00u) ? 1u : 0u;
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < ND; ++j)
                    if (j == i) {
                        // operand j's words and rank entries; the request slots of operands a narrower query lacks repeat its last one
                        if (lane == j || (j + 1u == ND && lane > j && lane < 3u)) {
                            rq_base = (unsigned long long)(uintptr_t)d.bitmap;
                            rq_mul = kGTW * 4u;
                        }
                        if (lane == 3u + j || (j + 1u == ND && lane > 3u + j && lane < 6u)) {
                            rq_base = (unsigned long long)(uintptr_t)d.rank_dir;
                            rq_mul = kGRk * 4u;
                        }
                        if (lane == 0) {
                            const uint16_t mr = d.max_raw;
                            sh[kShTs + j] = __float_as_uint(d.term_score);
                            sh[kShVmax + j] = (d.term_score > 0.0f && mr < 0x7C00u) ? __float_as_uint(posting_value(d.term_score, mr)) : 0x7F800000u;  // +inf: no bound
                            reinterpret_cast<unsigned long long*>(sh + kShScores)[j] = (unsigned long long)(uintptr_t)d.scores;
                        }
                    }
                role_of[k] = 1u + i;
                ++i;
            }
        }
        if (lane == 6u) {
            rq_base = (unsigned long long)(uintptr_t)cdocs;
            rq_mul = 16u;
        }
        if (lane == 7u) {
            rq_base = (unsigned long long)(uintptr_t)cscores;
            rq_mul = 8u;
        }
        const KOp root(gops + n);
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t k = root.and_order(j);
            if (lane == 0) sh[kShSrc + j] = k == 0u ? role_of[0] : k == 1u ? role_of[1] : k == 2u ? role_of[2] : role_of[3];
        }
    }
    const uint32_t n_spans = H->n_spans;
    const unsigned long long range = (unsigned long long)(H->doc_hi - H->doc_lo);
    const uint32_t span_lo = span == 0 ? H->doc_lo : ((H->doc_lo + (uint32_t)(range * span / n_spans)) & ~(kGT - 1u));
    const uint32_t span_hi = span + 1 == n_spans ? H->doc_hi : ((H->doc_lo + (uint32_t)(range * (span + 1) / n_spans)) & ~(kGT - 1u));
    const uint32_t bitmap_base = H->bitmap_base;
    const uint32_t t_first = (span_lo - bitmap_base) >> kProbeTileShift;
    const uint32_t t_end = span_hi > span_lo ? ((span_hi - 1u - bitmap_base) >> kProbeTileShift) + 1u : t_first;  // one behind the last tile
    if (lane < 8u) {
        reinterpret_cast<unsigned long long*>(ctx + kCRqBase)[lane] = rq_base;
        ctx[kCRqMul + lane] = rq_mul;
    }
    if (lane == 0) {
        auto put64 = [&](uint32_t w, unsigned long long v) { *reinterpret_cast<unsigned long long*>(ctx + w) = v; };
        put64(0, (unsigned long long)(uintptr_t)blob);
        put64(kCGthr, (unsigned long long)(uintptr_t)gthr);
        put64(kCPool, H->off_pool ? (unsigned long long)(uintptr_t)(blob + H->off_pool) : 0ull);
        put64(kCDocs, (unsigned long long)(uintptr_t)cdocs);
        put64(kCScores, (unsigned long long)(uintptr_t)cscores);
        put64(kCDir, (unsigned long long)(uintptr_t)ctdir);
        put64(kCUpper, H->key_upper);
        ctx[kCLen] = clen;
        ctx[kCTFirst] = t_first;
        ctx[kCTEnd] = t_end;
        ctx[kCBase] = bitmap_base;
        ctx[kCTopK] = H->top_k;
        ctx[kCStat] = H->stat_off;
        ctx[kCQ] = q;
        ctx[kCSpan] = span;
        ctx[kCKeys] = H->keys_base;
        const unsigned long long t0 = __hip_atomic_load(gthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // what other spans of the query have already reached
        *reinterpret_cast<unsigned long long*>(misc) = t0;
        misc[2] = 0u;
        misc[4] = 0u;
        misc[6] = ~(uint32_t)(t0 >> 32);  // (forces the first raw_min)
    }
    probe_lds_fence();
    (void)ring_refresh_raw_min<ND>(my);
}

// ------------------------------------------------------------------------------------------------ consumer: the tile loop of a span
template <uint32_t ND, uint32_t MAXND>
__device__ __noinline__ void ring_tiles(uint32_t* my_g, const uint32_t ctl, const uint32_t c, const uint32_t S, unsigned long long* __restrict__ span_keys, unsigned long long* __restrict__ num_hits) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t my = lds_addr(my_g);  // byte address of the consumer's block
    const uint32_t ctx = my + kOCtx * 4u, misc = my + kOMisc * 4u, sh = my + kOShape * 4u, uq = my + kOU * 4u, rq = my + kOR * 4u;
    const uint32_t slots = my + ring_priv_words(MAXND) * 4u;
    constexpr uint32_t kSlotBytes = ring_slot_words(MAXND) * 4u;
    constexpr uint32_t kIdsOff = MAXND * (kGTW + kGRk) * 4u, kScsOff = kIdsOff + kRingRounds * 1024u, kRankOff = MAXND * kGTW * 4u;
    RS_INIT

    // ---- what the loop keeps in registers
    uint32_t n_req = uni(lds_ld(ctx + kCNReq * 4u)), n_done = uni(lds_ld(ctx + kCNDone * 4u)), s_req = uni(lds_ld(ctx + kCSReq * 4u)), s_done = uni(lds_ld(ctx + kCSDone * 4u));
    const uint32_t t_end = uni(lds_ld(ctx + kCTEnd * 4u)), bitmap_base = uni(lds_ld(ctx + kCBase * 4u));
    uint32_t t_req = uni(lds_ld(ctx + kCTFirst * 4u));
    const bool stat = uni(lds_ld(ctx + kCStat * 4u)) != 0u;
    uint32_t raw_min = uni(lds_ld(misc + 5u * 4u));
    uint32_t thr_hi = uni(lds_ld(misc + 6u * 4u));  // the score half of the threshold raw_min belongs to
    const u32x2 rqb = *(const L2x32*)(ctx + (kCRqBase + 2u * (lane & 7u)) * 4u);  // lane j < 8: pointer j of a request
    const uint32_t rqm = *L(ctx + (kCRqMul + (lane & 7u)) * 4u);
    const unsigned long long rq_base = ((unsigned long long)rqb.y << 32) | rqb.x;
    const unsigned long long gthr_p = lds_ptr(ctx + kCGthr * 4u);
    unsigned long long* const gthr = reinterpret_cast<unsigned long long*>((uintptr_t)gthr_p);
    bool failed = false;

    // a slice of the cover's tile directory rides in a register (lane l: postings below tile dir_base + l)
    uint32_t dir_base = t_req;
    auto load_dir = [&]() {
        const uint32_t* ctdir = reinterpret_cast<const uint32_t*>((uintptr_t)lds_ptr(ctx + kCDir * 4u));
        return as_global(ctdir)[(dir_base + lane < t_end ? dir_base + lane : t_end) << kGDir];  // (entry t_end exists: one behind the last tile)
    };
    uint32_t dirv = load_dir();
    uint32_t h_t = 0, h_e0 = 0, h_e1 = 0;  // lane (n & 7): tile and cover slice of this consumer's n-th request (what the wave needs when the tile has landed)
    // ---- requests: the next non-empty tiles of the span, at most S ahead of the tile being worked on
    auto request_more = [&]() {
        while (n_req - n_done < S && t_req < t_end) {  // uniform
            if (t_req + 1u - dir_base >= 64u) {         // the directory slice is used up
                dir_base = t_req;
                dirv = load_dir();
            }
            const uint32_t e0 = rdl(dirv, t_req - dir_base), e1 = rdl(dirv, t_req + 1u - dir_base);
            if (e1 > e0) {  // uniform: a tile without cover postings has no hits — nothing of it is read
                const uint32_t v0 = (e0 >> 3) << 1;  // (the slice starts at a multiple of 8 postings: its scores at a multiple of 16 bytes)
                const uint32_t x = lane < 6u ? t_req : v0;
                const unsigned long long p = rq_base + (unsigned long long)x * rqm;
                const uint32_t r = my + s_req * (kReqWords * 4u);
                if (lane < 8u) *(L2x32*)(r + lane * 8u) = u32x2{(uint32_t)p, (uint32_t)(p >> 32)};
                if (lane == 8u) *L(r + 19u * 4u) = ((e1 - 1u) >> 2) - v0 + 1u;  // valid 16-byte vectors of the slice
                if (lane == (n_req & 7u)) {
                    h_t = t_req;
                    h_e0 = e0;
                    h_e1 = e1;
                }
                probe_lds_fence();
                ++n_req;
                s_req = s_req + 1u == S ? 0u : s_req + 1u;
                if (lane == 0) lds_st(ctl + c * 4u, n_req);
            }
            ++t_req;
        }
    };
    request_more();
    RS_AT(0)

    uint32_t un = 0, rn = 0, rhead = 0;
    unsigned long long hits = 0;
    unsigned long long g_prev = 0ull;
    uint32_t tile_lo = 0, tile = 0, rank = 0;  // the slot being worked on (byte addresses of words[MAXND][kGTW], rank[MAXND][kGRk])
    // ---- scoring of the ranked queue, 64 hits at a time, WITHOUT waiting for its gathers: a flush is a little pipeline that advances one
    // stage per service (k_scan_probe has the reasoning).  Stage i has the gather of dense operand i's scores in flight.
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
    uint32_t svcs_since_merge = 4;  // pool merges are spaced out: each is a round trip to memory under a lock
    auto after_push = [&](const uint32_t rm) {
        raw_min = uni(rm);
        thr_hi = uni(lds_ld(misc + 6u * 4u));
    };
    // One call per service: the flush in flight advances by one stage, or a new flush starts.  Exactly ONE gather goes out, at one place.
    auto flush_service = [&](const bool final) {
        unsigned long long gp = lds_ptr(sh + kShScores * 4u);
        uint32_t gidx = 0u;
        if (f_stage) {  // uniform: one more operand's value is known
            ProbeShape<ND> Sh;
            Sh.cts = __uint_as_float(*L(sh + kShCts * 4u));
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i) {
                Sh.ts[i] = __uint_as_float(*L(sh + (kShTs + i) * 4u));
                Sh.vmax[i] = __uint_as_float(*L(sh + (kShVmax + i) * 4u));
            }
#pragma unroll
            for (uint32_t j = 0; j <= ND; ++j) Sh.src[j] = *L(sh + (kShSrc + j) * 4u);
            Sh.prunable = true;
            const float thr_f = __uint_as_float(unorder_f32(lds_ld(misc + 4u)));  // NaN while there is no threshold: nothing is dropped
            bool last = true;
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i)
                if (f_stage == i + 1u) {  // uniform
                    f_vd[i] = posting_value(Sh.ts[i], f_r);
                    if (i + 1u < ND) {  // hits that can no longer reach the threshold drop out, the next operand's gather goes out
                        constexpr uint32_t zero = 0;
                        f_alive = f_alive && !(probe_sum<ND>(Sh, f_vc, f_vd) < thr_f);
                        if (stat && lane == 0) *L(misc + 16u) += 2u * (uint32_t)__popcll(wballot(f_alive));
                        gp = lds_ptr(sh + (kShScores + 2u * (i + 1u < ND ? i + 1u : zero)) * 4u);
                        gidx = f_alive ? f_idx[i + 1u < ND ? i + 1u : zero] : 0u;
                        last = false;
                    }
                }
            if (last) {  // the scores are final: the keys go to the candidate buffer
                const float score = probe_sum<ND>(Sh, f_vc, f_vd);
                const unsigned long long key = ((unsigned long long)order_f32(__float_as_uint(score)) << 32) | (unsigned long long)f_doc;
                f_stage = 0;
                const unsigned long long thr = *LL(misc), upper = *LL(ctx + kCUpper * 4u);
                const bool pending = f_alive && key > thr && key < upper;
                if (wballot(pending)) {  // uniform; rare once the threshold has warmed up
                    const bool merge = svcs_since_merge >= 4u || final;
                    if (merge) svcs_since_merge = 0;
                    after_push(ring_push_keys<ND, MAXND>(my_g, key, pending, merge));
                }
            } else ++f_stage;
        } else if (rn >= 64u || (final && rn)) {  // uniform: start a flush — the first operand's gather goes out
            const uint32_t count = rn < 64u ? rn : 64u;
            probe_lds_fence();
            f_alive = lane < count;
            uint32_t raw = 0;
            const uint32_t slot = ((rhead + lane) & (kRR - 1u)) * 4u;
            if (f_alive) {
                f_doc = *L(rq + slot);
                raw = *L(rq + kRR * 4u + slot);
#pragma unroll
                for (uint32_t i = 0; i < ND; ++i) f_idx[i] = *L(rq + (2u + i) * kRR * 4u + slot);
            }
            f_vc = posting_value(__uint_as_float(*L(sh + kShCts * 4u)), (uint16_t)raw);
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i) f_vd[i] = __uint_as_float(*L(sh + (kShVmax + i) * 4u));
            if (stat && lane == 0) *L(misc + 16u) += 2u * count;  // gathered bytes of the span
            gidx = f_alive ? f_idx[0] : 0u;
            f_stage = 1u;
            rhead = (rhead + count) & (kRR - 1u);
            rn -= count;
        }
        f_r = as_global(reinterpret_cast<const uint16_t*>((uintptr_t)gp))[gidx];
    };
    // Rank the first `cnt` (<= 64) entries of the unranked queue — live hits of the CURRENT tile, whose words are in its slot — into the
    // ranked queue: index in dense operand i = rank directory entry of the doc's 512-doc group + set bits of the group below the doc.
    auto rank_some = [&](const uint32_t cnt) {
        while (rn + cnt > kRR) {  // uniform: the ranked queue is full while a flush is still in flight (warm-up, a dense stretch of hits)
            after_push(ring_flush_sync<ND, MAXND>(my_g, rhead));
            rhead = (rhead + 64u) & (kRR - 1u);
            rn -= 64u;
        }
        probe_lds_fence();
        if (ND == 2u && cnt <= 32u) {  // uniform: the usual case — both operands at once, lanes 0-31 rank in operand 0, lanes 32-63 in operand 1
            const uint32_t el = lane & 31u, role = lane >> 5;
            if (el < cnt) {
                const uint32_t e = *L(uq + el * 4u);
                const uint32_t rel = e >> 16;  // doc - tile_lo
                const uint32_t slot = ((rhead + rn + el) & (kRR - 1u)) * 4u;
                if (role == 0u) {
                    *L(rq + slot) = tile_lo + rel;
                    *L(rq + kRR * 4u + slot) = e & 0xFFFFu;
                }
                const uint32_t g = rel >> kRankShift, wi = (rel >> 5) & 15u, below = (1u << (rel & 31u)) - 1u;
                const int full = (int)((1u << wi) - 1u);  // bit j: word j of the group lies entirely below the doc
                const uint32_t tl = tile + role * (kGTW * 4u);
                uint32_t acc = *L(rank + (role * kGRk + g) * 4u) + (uint32_t)__popc(*L(tl + (rel >> 5) * 4u) & below);
                const L4x32* gw = (const L4x32*)(tl + g * 64u);
#pragma unroll
                for (uint32_t v4 = 0; v4 < 4; ++v4) {
                    const u32x4 x = gw[v4];
                    acc += (uint32_t)__popc(x.x & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 0u, 1u));
                    acc += (uint32_t)__popc(x.y & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 1u, 1u));
                    acc += (uint32_t)__popc(x.z & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 2u, 1u));
                    acc += (uint32_t)__popc(x.w & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 3u, 1u));
                }
                *L(rq + (2u + role) * kRR * 4u + slot) = acc;
            }
        } else if (lane < cnt) {
            const uint32_t e = *L(uq + lane * 4u);
            const uint32_t rel = e >> 16;  // doc - tile_lo
            const uint32_t slot = ((rhead + rn + lane) & (kRR - 1u)) * 4u;
            *L(rq + slot) = tile_lo + rel;
            *L(rq + kRR * 4u + slot) = e & 0xFFFFu;
            const uint32_t g = rel >> kRankShift, wi = (rel >> 5) & 15u, below = (1u << (rel & 31u)) - 1u;
            const int full = (int)((1u << wi) - 1u);
#pragma unroll
            for (uint32_t i = 0; i < ND; ++i) {
                const uint32_t tl = tile + i * (kGTW * 4u);
                uint32_t acc = *L(rank + (i * kGRk + g) * 4u) + (uint32_t)__popc(*L(tl + (rel >> 5) * 4u) & below);
                const L4x32* gw = (const L4x32*)(tl + g * 64u);
#pragma unroll
                for (uint32_t v4 = 0; v4 < 4; ++v4) {
                    const u32x4 x = gw[v4];
                    acc += (uint32_t)__popc(x.x & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 0u, 1u));
                    acc += (uint32_t)__popc(x.y & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 1u, 1u));
                    acc += (uint32_t)__popc(x.z & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 2u, 1u));
                    acc += (uint32_t)__popc(x.w & (uint32_t)__builtin_amdgcn_sbfe(full, v4 * 4u + 3u, 1u));
                }
                *L(rq + (2u + i) * kRR * 4u + slot) = acc;
            }
        }
        rn += cnt;
        if (un > cnt) {  // uniform
            const uint32_t rem = un - cnt;
            constexpr uint32_t kMove = kRU / 64u;
            uint32_t t[kMove];
#pragma unroll
            for (uint32_t r = 0; r < kMove; ++r) t[r] = r * 64u + lane < rem ? *L(uq + (cnt + r * 64u + lane) * 4u) : 0u;
            probe_lds_fence();
#pragma unroll
            for (uint32_t r = 0; r < kMove; ++r)
                if (r * 64u + lane < rem) *L(uq + (r * 64u + lane) * 4u) = t[r];
        }
        un -= cnt;
    };
    // one round of 256 cover postings (lane l: four consecutive ones, the first of them posting `p0` of the list) against the tile's words
    uint32_t e_lo = 0, e_cnt = 0;  // the cover's postings [e_lo, e_lo + e_cnt) belong to the tile
    struct ProbeWords {
        uint32_t w[4];
    };
    auto probe_read = [&](const u32x4 d4) {  // the operands' words at the four postings of a lane (AND of the operands)
        const uint32_t dd[4] = {d4.x, d4.y, d4.z, d4.w};
        ProbeWords pw;
#pragma unroll
        for (uint32_t cc = 0; cc < 4; ++cc) {
            const uint32_t a = tile + ((dd[cc] >> 3) & ((kGTW - 1u) << 2));  // byte address of the doc's word
            pw.w[cc] = *L(a);
#pragma unroll
            for (uint32_t i = 1; i < ND; ++i) pw.w[cc] &= *L(a + i * (kGTW * 4u));
        }
        return pw;
    };
    auto probe_eval = [&](const u32x4 d4, const u32x2 s2, const ProbeWords& pw, const uint32_t p0) {
        const uint32_t dd[4] = {d4.x, d4.y, d4.z, d4.w};
        const uint32_t rw[4] = {s2.x & 0xFFFFu, s2.x >> 16, s2.y & 0xFFFFu, s2.y >> 16};
        const uint32_t x0 = p0 - e_lo;
        unsigned long long lm[4];
        bool live[4];
#pragma unroll
        for (uint32_t cc = 0; cc < 4; ++cc) {
            const bool in = (x0 + cc) < e_cnt;  // a posting of this tile (lanes behind the slice hold copies of its last vector)
            const bool bit = __builtin_amdgcn_ubfe(pw.w[cc], dd[cc], 1u) != 0u;  // (the bit field's offset is taken modulo 32)
            const bool strong = rw[cc] >= raw_min;
            const unsigned long long sm = wballot(in) & wballot(bit);
            hits += (unsigned long long)__popcll(sm);
            lm[cc] = sm & wballot(strong);
            live[cc] = in && bit && strong;
        }
        if (lm[0] | lm[1] | lm[2] | lm[3]) {  // uniform
#pragma unroll
            for (uint32_t cc = 0; cc < 4; ++cc) {
                if (live[cc]) *L(uq + (un + (uint32_t)__popcll(lm[cc] & ((1ull << lane) - 1ull))) * 4u) = ((dd[cc] - tile_lo) << 16) | rw[cc];
                un += (uint32_t)__popcll(lm[cc]);
            }
        }
    };

    uint32_t svc = 0;  // tiles since the wave last looked at what it has in flight
    while (n_done != n_req && !failed) {  // uniform: tiles asked for and not yet worked on
        // ---- every kRingSvc-th tile: everything this wave has in flight (the flush's gather, the threshold word) is waited for, the flush
        // pipeline advances a stage, the query's threshold word is asked for again.  A round trip to memory under the stream's load takes
        // longer than a tile: waiting at EVERY tile made the wave's own loads its critical path.
        if (svc == 0u) {  // uniform
            __builtin_amdgcn_s_waitcnt(0x0F70);
            {
                uint32_t fr = f_r, glo = (uint32_t)g_prev, ghi = (uint32_t)(g_prev >> 32);
                asm volatile("" : "+v"(fr), "+v"(glo), "+v"(ghi), "+v"(dirv));
                f_r = (uint16_t)fr;
                g_prev = ((unsigned long long)ghi << 32) | glo;
            }
            if (lane == 0 && g_prev > *LL(misc)) *LL(misc) = g_prev;  // what other spans of the query have published (QHeader::gthr)
            ++svcs_since_merge;
            flush_service(false);
            if (lane == 0) g_prev = __hip_atomic_load(gthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            probe_lds_fence();
            if (uni(lds_ld(misc + 4u)) != thr_hi) after_push(ring_refresh_raw_min<ND>(my_g));  // uniform: the threshold's score has moved
        }
        svc = svc + 1u == kRingSvc ? 0u : svc + 1u;
        RS_AT(1)
        RS_COUNT(8)
        // ---- the tile's slot: its postings are read together with the loader's count (a slot is usually there: the consumers are the
        // slower side); if the tile has not been published yet, wait and read again
        const uint32_t slot = slots + s_done * kSlotBytes;
        const uint32_t hl = n_done & 7u;
        const uint32_t t = rdl(h_t, hl), e0 = rdl(h_e0, hl), e1 = rdl(h_e1, hl);
        const uint32_t v0 = (e0 >> 3) << 1;  // first 16-byte vector of the slice
        const uint32_t rounds = (e1 - v0 * 4u + 255u) >> 8;
        u32x4 d4[kRingRounds];
        u32x2 s2[kRingRounds];
        {
            uint32_t fc = lds_ld(ctl + (16u + c) * 4u);
#pragma unroll
            for (uint32_t r = 0; r < kRingRounds; ++r) {
                d4[r] = *(const L4x32*)(slot + kIdsOff + (r * 64u + lane) * 16u);
                s2[r] = *(const L2x32*)(slot + kScsOff + (r * 64u + lane) * 8u);
            }
            probe_lds_fence();
            if ((int32_t)(uni(fc) - n_done) <= 0) {  // uniform: not there yet
                RS_COUNT(9)
                uint32_t spin = 0;
                do {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spin > kRingSpin) {
                        failed = true;
                        break;
                    }
                    fc = lds_ld(ctl + (16u + c) * 4u);
                    probe_lds_fence();
                } while ((int32_t)(uni(fc) - n_done) <= 0);
                if (failed) break;
#pragma unroll
                for (uint32_t r = 0; r < kRingRounds; ++r) {
                    d4[r] = *(const volatile L4x32*)(slot + kIdsOff + (r * 64u + lane) * 16u);
                    s2[r] = *(const volatile L2x32*)(slot + kScsOff + (r * 64u + lane) * 8u);
                }
                probe_lds_fence();
            }
        }
        RS_AT(2)
        tile = slot;
        rank = slot + kRankOff;
        tile_lo = bitmap_base + (t << kProbeTileShift);
        e_lo = e0;
        e_cnt = e1 - e0;
#ifndef VQ_RING_STREAM_ONLY  // (diagnostic build: slots are handed back unread — what the loader side alone takes)
        {  // the slot's rounds: every LDS read of the tile's words goes out first, then the few hits are picked up
            ProbeWords pw[kRingRounds];
#pragma unroll
            for (uint32_t r = 0; r < kRingRounds; ++r)
                if (r < rounds) pw[r] = probe_read(d4[r]);  // uniform
#pragma unroll
            for (uint32_t r = 0; r < kRingRounds; ++r)
                if (r < rounds) {  // uniform
                    while (r && un > kRU - 256u) rank_some(un < 64u ? un : 64u);  // uniform: room for another round
                    probe_eval(d4[r], s2[r], pw[r], (v0 + r * 64u + lane) * 4u);
                }
        }
        if (rounds > kRingRounds) {  // uniform: a dense stretch of the cover — further rounds are fetched on the spot
            const VQ_GLOBAL u32x4* cd4 = as_global(reinterpret_cast<const u32x4*>((uintptr_t)lds_ptr(ctx + kCDocs * 4u)));
            const VQ_GLOBAL u32x2* cs2 = as_global(reinterpret_cast<const u32x2*>((uintptr_t)lds_ptr(ctx + kCScores * 4u)));
            const uint32_t nvec = (uni(lds_ld(ctx + kCLen * 4u)) + 3u) >> 2;
            for (uint32_t r = kRingRounds; r < rounds; ++r) {
                while (un > kRU - 256u) rank_some(un < 64u ? un : 64u);  // uniform: room for another round
                const uint32_t v = v0 + r * 64u + lane;
                u32x4 dx = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                u32x2 sx = u32x2{0u, 0u};
                if (v < nvec) {
                    dx = cd4[v];
                    sx = cs2[v];
                }
                const ProbeWords pw = probe_read(dx);
                probe_eval(dx, sx, pw, v * 4u);
            }
        }
        RS_AT(3)
        if (un) RS_COUNT(10)
        while (un) rank_some(un < 64u ? un : 64u);  // uniform: the tile's live hits are ranked while its words are still in the slot
#else
        hits += (unsigned long long)__popcll(wballot((d4[0].x ^ s2[0].x) == 0x12345u && rounds == 0xFFFFFFFFu));
#endif
        probe_lds_fence();  // every read of the slot has returned: the slot may be asked for again
        ++n_done;
        s_done = s_done + 1u == S ? 0u : s_done + 1u;
        RS_AT(4)
        request_more();
        RS_AT(5)
    }
    if (svc) {  // the flush's gather and the threshold word may still be in flight
        __builtin_amdgcn_s_waitcnt(0x0F70);
        uint32_t fr = f_r;
        asm volatile("" : "+v"(fr));
        f_r = (uint16_t)fr;
    }
    while (f_stage || rn) flush_service(true);  // uniform: the flush pipeline drains (these gathers are waited for where they are used)
    // ---- the span's end: its best keys, its hit count
    {
        uint32_t* const myp = my_g;
        const CandState cs = ring_cand_state<MAXND>(myp);
        const uint32_t top_k = myp[kOCtx + kCTopK];
        ring_cand_prune(cs, top_k, lane);
        const uint32_t cn = *cs.n;
        unsigned long long* out = span_keys + (size_t)myp[kOCtx + kCKeys] + (size_t)myp[kOCtx + kCSpan] * top_k;
        for (uint32_t i = lane; i < top_k; i += 64u) out[i] = i < cn ? cs.cand[i] : 0ull;
        if (failed) hits |= 1ull << 60;  // a launch that gave up on a hand-off must not look like a result
        const uint32_t q = myp[kOCtx + kCQ];
        if (lane == 0 && hits) atomicAdd(&num_hits[q], hits);
        if (lane == 0 && stat && myp[kOMisc + 4]) atomicAdd(&num_hits[myp[kOCtx + kCStat]], (unsigned long long)myp[kOMisc + 4]);
        if (lane == 0) {
            myp[kOCtx + kCNReq] = n_req;
            myp[kOCtx + kCNDone] = n_done;
            myp[kOCtx + kCSReq] = s_req;
            myp[kOCtx + kCSDone] = s_done;
            myp[kOCtx + kCFailed] = failed ? 1u : 0u;
        }
        probe_lds_fence();
    }
    RS_AT(6)
    RS_COUNT(11)
    RS_FLUSH(0, 16)
}

// Work items are (query, span) pairs in ROUND-major order — span 0 of every query first, then span 1, ... — so that a query's later spans
// start with the threshold its first ones have already put into the pool.  spans_each != 0: every query has that many spans
// (item i = query i % nq, span i / nq); otherwise items[i] = query << 12 | span.
template <uint32_t MAXND, uint32_t C>
__global__ __launch_bounds__((kRingL + C) * 64) void k_scan_ring(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ blob_off, const uint32_t* __restrict__ qmap,
                                                                 const uint32_t nq, const uint32_t spans_each, const uint32_t* __restrict__ items, const uint32_t total_items,
                                                                 uint32_t* __restrict__ work, const uint32_t S, unsigned long long* __restrict__ span_keys,
                                                                 unsigned long long* __restrict__ num_hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < kRCtl) lds[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t* const err = work + 1;
    const uint32_t ctl = uni(lds_addr(lds));
    if (wave < kRingL) {
        ring_loader<MAXND, C>(ctl, S, wave, err);
        return;
    }
    const uint32_t c = wave - kRingL;
    uint32_t* const my = lds + kRCtl + c * ring_cons_words(MAXND, S);
    if (lane == 0) {
        my[kOCtx + kCNReq] = 0u;
        my[kOCtx + kCNDone] = 0u;
        my[kOCtx + kCSReq] = 0u;
        my[kOCtx + kCSDone] = 0u;
        my[kOCtx + kCFailed] = 0u;
    }
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(work, 1u);
    item = uni(item);
    bool failed = false;
    while (item < total_items && !failed) {  // uniform
        uint32_t nxt = 0;
        if (lane == 0) nxt = atomicAdd(work, 1u);  // (in flight while this item is worked on)
        uint32_t ql, span;
        if (spans_each) {
            span = item / nq;
            ql = item - span * nq;
        } else {
            const uint32_t it = as_const<uint32_t>(items)[item];
            ql = it >> 12;
            span = it & 0xFFFu;
        }
        const uint32_t q = as_const<uint32_t>(qmap)[ql];
        const uint8_t* blob = blobs + as_const<uint32_t>(blob_off)[q];
        const uint32_t n = as_const<QHeader>(blob)->simple_n;
        if (n == 2u) {
            ring_setup<1>(blob, span, q, my);
            ring_tiles<1, MAXND>(my, ctl, c, S, span_keys, num_hits);
        } else if (MAXND >= 2u && n == 3u) {
            ring_setup<(MAXND >= 2u ? 2u : 1u)>(blob, span, q, my);
            ring_tiles<(MAXND >= 2u ? 2u : 1u), MAXND>(my, ctl, c, S, span_keys, num_hits);
        } else if (MAXND >= 3u) {
            ring_setup<(MAXND >= 3u ? 3u : 1u)>(blob, span, q, my);
            ring_tiles<(MAXND >= 3u ? 3u : 1u), MAXND>(my, ctl, c, S, span_keys, num_hits);
        }
        failed = uni(lds_ld(lds_addr(my + kOCtx + kCFailed))) != 0u;
        item = uni(nxt);
    }
    if (failed && lane == 0) atomicAdd(err, 1u);
    if (lane == 0) lds_st(ctl + (32u + c) * 4u, 1u);
}

template <uint32_t MAXND, uint32_t C>
static void launch_ring_t(hipStream_t st, uint32_t grid, uint32_t S, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* qmap, uint32_t nq, uint32_t spans_each,
                          const uint32_t* items, uint32_t total_items, uint32_t* work, unsigned long long* span_keys, unsigned long long* num_hits) {
    const size_t lds = scan_ring_lds_bytes(MAXND, C, S);
    static bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan_ring<MAXND, C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr;
    hipLaunchKernelGGL((k_scan_ring<MAXND, C>), dim3(grid), dim3((kRingL + C) * 64), lds, st, blobs, blob_off, qmap, nq, spans_each, items, total_items, work, S, span_keys, num_hits);
}

uint32_t scan_ring_consumers(uint32_t max_nd) {  // consumer waves per workgroup: as many as leave every consumer two slots
    return max_nd <= 2u ? 6u : 4u;
}

// max_nd: most dense operands of a query of the launch (sizes the slots); work: two zeroed u32 (the item counter, the error word)
void launch_scan_ring(hipStream_t st, uint32_t max_nd, uint32_t grid, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* qmap, uint32_t nq, uint32_t spans_each,
                      const uint32_t* items, uint32_t total_items, uint32_t* work, unsigned long long* span_keys, unsigned long long* num_hits) {
    if (!total_items) return;
    const uint32_t C = scan_ring_consumers(max_nd);
    const uint32_t S = scan_ring_slots(max_nd, C);
#define VQ_RING_LAUNCH(ND, CC) launch_ring_t<ND, CC>(st, grid, S, blobs, blob_off, qmap, nq, spans_each, items, total_items, work, span_keys, num_hits)
    if (max_nd <= 1u) VQ_RING_LAUNCH(1, 6);
    else if (max_nd == 2u) VQ_RING_LAUNCH(2, 6);
    else VQ_RING_LAUNCH(3, 4);
#undef VQ_RING_LAUNCH
}

}  // namespace vq
