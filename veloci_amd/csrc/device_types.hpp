// Device-visible descriptors of one compiled query ("query blob") — shared by the host compiler
// (compile.cpp) and the kernels (kernels.hip).  All offsets are bytes from the blob start; every
// section is 8-byte aligned.
//
// A query is evaluated tile by tile over the shard's doc-id space:
//   lists   — sorted, unique doc-id lists living in HBM: posting lists (doc u32 + f16 score),
//             id-only lists (phrase-pair anchors, text_id_to_anchor rows, tokens_to_text_id rows,
//             small host-built lists carried inside the blob)
//   ops     — postfix program of the score tree (reference Request.search_req)
//   fops    — postfix program of the filter tree (presence only)
//   groups / tboosts / cols / locf / facets — the per-hit sink stages, in the order the reference
//             applies them (plan_creator/execution_plan.rs:163-193, search.rs:176-206)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vq {

constexpr int kBlock = 64;          // threads per workgroup: ONE wave (no cross-wave barriers anywhere in the scan)
constexpr int kCandCap = 2048;      // LDS candidate buffer (u64 keys) per workgroup
constexpr int kMaxTopK = 1024;      // top + skip supported in-kernel
constexpr int kStackDepth = 16;     // deepest postfix evaluation stack (LDS is sized to the batch's real depth)
constexpr int kMaxChildren = 16;    // children per AND/OR node (the query generator ORs one leaf per term and field)
constexpr int kMaxLists = 64;       // lists per query in one launch
constexpr int kMaxOps = 160;
constexpr int kMaxSkipWhen = 4;
constexpr uint32_t kTileDirShift = 14;  // tile directory of an id list: one entry per 16384 docs (the tile of k_scan_ring)
constexpr uint32_t kProbeTileShift = 15;  // the tile of k_scan_probe: 32768 docs = every second directory entry
constexpr uint32_t kRankShift = 9;  // rank directory of a dense list: one entry per 512 docs (16 bitmap words = one 64-byte sector)

enum ListFlags : uint32_t {
    LIST_HAS_SCORES = 1u,  // posting list: f16 anchor scores, value = term_score * (f16 / 100)
    LIST_COVER = 2u,       // part of the cover set that decides which tiles are visited
    LIST_BITMAP = 4u,      // a bitmap image of the list exists in HBM (dense lists): tiles are copied, not scattered
    LIST_F32 = 8u,         // materialised leaf (k_union): scores are final f32 values, term_score is not applied
};

struct DList {  // 56 B
    const uint32_t* docs;    // 16-byte aligned, padded to a multiple of 4 with 0xFFFFFFFF
    const uint16_t* scores;  // f16 bits, same indexing as docs (null for id-only lists)
    uint32_t len;
    uint32_t flags;
    float term_score;        // s_t (search_field.rs:426), request.boost folded in (:359-364)
    uint16_t max_raw;        // largest f16 bit pattern among the list's scores (all non-negative): bounds every posting value of the list
    uint16_t pad;
    const uint32_t* bitmap;    // LIST_BITMAP: bit (doc - bitmap_base) set for every doc of the list
    const uint32_t* rank_dir;  // LIST_BITMAP: entries of the list below doc bitmap_base + (k << kRankShift)
    const uint32_t* tile_dir;  // lists of at least 1/4096 of the shard's docs (null otherwise): entries below doc bitmap_base + (k << kTileDirShift)
};

enum OpKind : uint8_t { OP_LEAF = 0, OP_AND = 1, OP_OR = 2, OP_BOOST1N = 3, OP_LEAFMAX = 4 };  // OP_LEAFMAX: only as a DSimple2 group kind — one leaf over several posting lists, its value the largest of the present ones (search_field.rs:455-461);  // OP_BOOST1N: unary, 1:n field boost of the leaf below (list_begin = anchors + f32 values, child_slot[0] = index into cols)

struct DOp {  // 40 B
    uint8_t kind;
    uint8_t nchild;
    uint8_t nslots;  // OR: number of distinct term slots (set_op.rs:122-124)
    uint8_t pad;
    uint16_t list_begin;  // LEAF: lists [list_begin, list_begin + list_count) — union, max score (search_field.rs:453-464)
    uint16_t list_count;
    uint8_t child_slot[kMaxChildren];  // OR: term slot of each child, children in stack (request) order
    uint8_t and_order[kMaxChildren];   // AND: child indices in summation order: others first, shortest last (set_op.rs:393,415-416)
};

// Presence program: the AND/OR/filter trees as three-address code over tile bitmaps (k_tile_scan P3).
// A slot reference is a list index, or (bit 15 set) a temporary bitmap; 0xFFFF = the root words.
enum PresKind : uint8_t { PRES_AND = 0, PRES_OR = 1, PRES_ZERO = 2, PRES_COUNT = 3 };  // PRES_COUNT: add popcount(in[0]) to counter `out` (count pre-pass)
constexpr uint16_t kSlotTemp = 0x8000;
constexpr uint16_t kSlotRoot = 0xFFFF;
struct DPresOp {  // 8 B
    uint8_t kind;
    uint8_t pad;
    uint16_t out;       // temp slot (kSlotTemp | t) or kSlotRoot
    uint16_t n_in;
    uint16_t in_begin;  // index into the slot reference array
};

struct DGroup {  // phrase group: present in ANY list -> multiply once (plan_steps.rs:235-277)
    uint16_t list_begin, list_count;
    float mult;
};

struct DTermBoost {  // one multiplication per list containing the doc (boost.rs:380-402)
    uint16_t list;
    uint16_t pad;
    float mult;
};

enum BoostFun : int32_t { BF_NONE = -1, BF_LOG2 = 0, BF_LOG10 = 1, BF_MULTIPLY = 2, BF_ADD = 3, BF_REPLACE = 4 };
enum ExprOp : int32_t { EX_NONE = -1, EX_DIV = 0, EX_MUL = 1, EX_ADD = 2, EX_SUB = 3 };

struct DColBoost {  // boost.rs:283-377, 470-504
    const float* values;      // dense column over [key_base, key_base + num_keys)
    const uint32_t* present;  // bitmap over the same keys (null = all present)
    uint32_t key_base, num_keys;
    int32_t fun;
    float param;
    uint32_t nskip;
    float skip[kMaxSkipWhen];
    int32_t expr_op;      // expression "x op y" (expression.rs:26-46); operand kind 0 = $SCORE (the boost value), 1 = constant
    int32_t expr_lkind, expr_rkind;
    float expr_lval, expr_rval;
    uint32_t pad;
};

constexpr uint16_t kLocPrecomputed = 0xFFFF;
struct DLocField {  // one text field with >= 2 query terms (boost.rs:34-87)
    uint16_t list_begin, list_count;  // identity column: loc_idx[list_begin .. + list_count) are the terms' token->text lists (any list of
                                      // the query, possibly the posting lists themselves), counted per doc;
                                      // list_count == kLocPrecomputed: list_begin is one (anchor, f32 2*c*c) list
};

struct DFacet {  // facet.rs:31-73 fast path: anchor -> value ids, counted into a histogram
    const uint64_t* offsets;  // CSR over [key_base, key_base + num_keys]
    const uint32_t* values;
    const uint32_t* direct;   // != null: no anchor has more than one value — direct[row] is the value id or 0xFFFFFFFF
    uint32_t key_base, num_keys;
    uint32_t hist_off;    // u32 index into the batch's histogram area
    uint32_t num_values;  // histogram length == dictionary size of the facet field
    uint32_t top;         // entries to report
    uint32_t out_off;     // index into the facet output arrays
};

// "Rich simple" queries (k_scan_simple<NV, true>): <= 4 single-list posting leaves in a tree of depth <= 2 (root over leaves or over
// AND / OR groups of leaves), plus the sink stages that only need membership in a few id lists ("side" lists: phrase groups,
// boost_term), the presence of the leaves themselves (text locality on shared lists) or a gather by doc id (column boosts).
struct DSimple2 {
    uint8_t ngroups, root_kind, root_nslots, n_side;
    uint8_t g_kind[4], g_mask[4], g_nslots[4];  // group g: OP_LEAF / OP_AND / OP_OR over the leaves in g_mask
    uint8_t g_order[4][4];                      // AND group: its leaves in summation order (set_op.rs:393,415-416)
    uint8_t g_slot[4][4];                       // OR group: term slot of leaf k
    uint8_t r_order[4];                         // root AND: groups in summation order
    uint8_t r_slot[4];                          // root OR: term slot of group g
    uint8_t n_grp, n_tb, n_loc, has_filter;
    uint8_t filter_mask, f32_mask, pad2, pad3;  // f32_mask: leaf k is a materialised list (LIST_F32: final f32 values); has_filter: a doc must be in one of these side lists (a filter that is one leaf)
    uint8_t grp_mask[4];                        // phrase group g: its side lists (bit s)
    uint8_t tb_side[4];                         // boost_term t: its side list
    uint8_t loc_leaf[2], loc_side[2];           // locality field f: leaves / side lists that are the terms' token->text lists
    uint16_t leaf_list[4];                      // list index of leaf k
    uint16_t side_list[4];                      // list index of side list s
    float grp_mult[4];
    float tb_mult[4];
    // Exact top-k pruning at queueing time (round 4): ub[m] bounds the score — score tree and column boosts, on the leaves' list maxima and the
    // boost columns' largest values — of a doc that holds exactly the leaves of mask m; the kernel multiplies in the phrase / term boosts and the
    // text-locality factor the doc really gets (its side-list memberships are known when it is queued) and drops the doc — counted as a hit,
    // never scored — when that product lies below the span's threshold.  prune == 0: some factor is not monotone (a negative boost, an expression,
    // skip_when_score, a boost column with values that turn a factor negative) or the request wants facets: every hit is scored.
    float ub[16];
    uint8_t prune, pad4, pad5, pad6;
};

// "Wide" queries (k_scan_wide): 5..16 single-list posting leaves in a tree of depth <= 2 — the query generator's shapes, one leaf per term
// and field (src/query_generator.rs:175-246): a flat OR / AND over the leaves, or a root over AND / OR groups of leaves.  No filter, no sink
// stages.  Leaves of a group are consecutive; a leaf directly under the root is a group of its own (g_kind = OP_LEAF).
constexpr int kWideMax = 16;
struct DWide {
    uint8_t n_leaves, n_groups, root_kind, root_nslots;
    uint8_t seq, pad0, pad1, pad2;                               // seq: tiles are visited in order (a dense list is in the cover)
    uint16_t bitmap_mask, cover_mask, prefetch_mask, f32_mask;   // bit k: leaf k is read as a bitmap image / is in the cover / prefetches its
                                                                 // next id vector / is a materialised list (LIST_F32)
    uint16_t leaf_list[kWideMax];                                // list index of leaf k
    uint8_t leaf_slot[kWideMax];                                 // OR group: term slot of leaf k (set_op.rs:122-124)
    uint8_t g_kind[kWideMax], g_begin[kWideMax], g_count[kWideMax];  // group g: OP_LEAF / OP_AND / OP_OR over leaves [g_begin, g_begin + g_count)
    uint8_t leaf_and_order[kWideMax];    // AND group g: entries [g_begin, g_begin + g_count) = its leaves in summation order (set_op.rs:393,415-416)
    uint8_t leaf_slot_order[kWideMax];   // OR group g: the same range = its leaves ordered by term slot
    uint8_t r_and_order[kWideMax];       // root AND: groups in summation order
    uint8_t r_slot[kWideMax];            // root OR: term slot of group g
    uint8_t r_slot_order[kWideMax];      // root OR: groups ordered by term slot
};

struct QHeader {
    uint32_t n_lists, n_ops, n_fops, n_groups, n_tboost, n_col, n_locf, n_facets;
    uint32_t off_lists, off_ops, off_fops, off_groups, off_tboost, off_col, off_locf, off_facets;
    uint32_t top_k;       // top + skip (search.rs:211)
    uint32_t tile_words;  // W / 32, power of two
    uint32_t n_spans;
    uint32_t keys_base;   // span s writes its top_k keys at span_keys[keys_base + s * top_k]
    uint32_t doc_lo, doc_hi;
    uint32_t part_keys_off;  // u64 index of this query's top_k keys inside the partial buffer's key area
    uint32_t blob_bytes;
    uint32_t desc_bytes;     // leading part of the blob that the kernel stages into LDS (everything but inline lists)
    uint32_t n_pres, off_pres, off_pres_in, n_temps;
    uint32_t off_loc_idx;    // u16 list indices referenced by the identity-column DLocFields
    uint32_t off_simple2;    // DSimple2 (simple_flags bit 18) or DWide (simple_flags bit 24)
    uint32_t off_pool;       // != 0: the query's shared top-k pool (DPool + top_k keys), written by the spans of k_scan_probe
    uint32_t prune_n;        // k_tile_scan top-k pruning: != 0: number of lists in prune_mask; a doc present in k of them scores at most
                             // unorder(prune_gbits[k]) (monotone in k), so docs with too few of them are counted as hits but never scored
    uint64_t prune_mask;     // the leaf lists of the score tree
    unsigned long long gthr; // written by the kernels: the best top-k threshold any span of this query has reached so far (atomicMax);
                             // every span may drop docs below it — some span alone already holds top_k better ones
    unsigned long long key_upper;  // only keys BELOW this enter the top-k (~0: no bound): page p of a deep request ranks what lies below the last key of page p-1
    uint32_t seq_tiles;      // k_tile_scan: a dense list is in the cover -> every tile of the span is visited, dense lists are copied from
                             // their bitmap images (their LIST_COVER flag is dropped)
    uint32_t stat_off;       // u64 index (relative to the `num_hits` base the kernels get) of this query's gathered-bytes counter: bytes the
                             // scan read by per-hit gathers (f16 scores 2 B, f32 values / boost columns / facet values 4 B, CSR offset pairs 16 B)
    uint32_t prune_gbits[16];
    uint32_t n_counts;       // != 0: count pre-pass — only the presence program runs, PRES_COUNT counters are added to
                             // counts[part_keys_off + c] (the buffer passed as `num_hits`); nothing is scored
    uint32_t bitmap_base;    // doc id of bit 0 of every list bitmap of this shard (multiple of 65536)
    uint32_t simple_n;       // != 0: the score tree is simple_n single-list posting leaves under one AND/OR (or a single leaf)
    uint32_t simple_flags;   // bits 0-3: leaf k is read as a bitmap; bits 8-11: leaf k is in the cover; bit 16: tiles are
                             // visited sequentially (a dense list is in the cover); bit 17: eligible for k_scan_simple;
                             // bit 18: rich simple query (DSimple2); bit 19: one materialised leaf (k_scan_leaf_f32); bits 20-23: leaf k has enough
                             // entries per tile to prefetch its next 1 KiB round; bit 24: wide query (DWide, k_scan_wide); bit 25: AND whose
                             // cover is ONE id list and whose other leaves are bitmap images or 16-bit arrays per tile (k_scan_probe, DProbe);
                             // bits 12-15: leaf k is probed as a 16-bit array; bit 28: a single leaf whose tile-packed image
                             // k_scan_union streams (DProbe::leaf[0]); bit 27: the root is an OR (k_scan_probe_or: docs without the cover
                             // are counted but not scored — CompiledQuery::or_skip_bound); bit 26: all operands bitmaps, top_k <= 32, opted in: k_scan_ring
};

// k_scan_probe's view of a leaf (simple_flags bit 25; at QHeader::off_simple2): the tile-packed image of its posting list — the list's
// postings grouped by 32768-doc tile, every tile padded to a multiple of 8 entries (a granule) with all-ones entries.
struct DProbeLeaf {
    const uint32_t* cov32;  // (doc - tile_lo) << 16 | f16 score: what the cover streams; an array operand's scores are gathered from here
    const uint16_t* arr16;  // doc - tile_lo, same indexing (null: some tile holds more than 2048 entries)
    const uint32_t* gdir;   // granules below tile k (tile 0 starts at QHeader::bitmap_base)
};
struct DProbe {
    DProbeLeaf leaf[4];
};

// The best top_k keys any span of the query has scored so far (k_scan_probe, top_k <= kPoolMaxK): a span merges its own best keys in under
// the lock and takes the pool's k-th key as its threshold — the k-th best of ALL spans' hits, which the largest of the spans' own k-th
// bests (QHeader::gthr alone) approaches only slowly.  Keys are unique (score bits, doc id), the merge drops duplicates.
constexpr uint32_t kPoolMaxK = 32;
struct DPool {
    uint32_t lock, n;
    // unsigned long long keys[top_k] follow (descending)
};

// Layout of the packed partial buffer (one per shard and batch; identical size on every shard):
//   [u64 num_hits[nq]] [u64 gathered_bytes[nq]] [u64 keys[total_keys]] [u32 hist[total_hist]]
// Everything in front of `off_hist` is exchanged by an all-gather; the histograms may be summed over the shards by an all-reduce
// instead (vq_partial_hist_*), SURVEY.md §8e.
struct PartialLayout {
    uint64_t nq, total_keys, total_hist;
    uint64_t off_hits, off_stats, off_keys, off_hist, bytes;
};

// 64-bit ranking key: (order-preserving f32 bits << 32) | doc  — larger == better under
// (score desc, id desc) (search.rs:122-130).
__host__ __device__ inline uint32_t order_f32(uint32_t bits) { return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u); }
__host__ __device__ inline uint32_t unorder_f32(uint32_t o) { return (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; }

}  // namespace vq
