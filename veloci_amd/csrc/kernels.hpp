// Host-callable launchers of kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_types.hpp"

namespace vq {

struct DictProbe {  // one fuzzy / prefix scan of a dictionary (k_dict_scan)
    uint32_t m, max_d, flags, lm;  // flags: 1 transposition costs one, 2 prefix (starts_with); lm: code points of `lquery`, or 0xFFFFFFFF
    uint16_t query[64];            // match side: the ORIGINAL term (lower-cased per code point when case-insensitive)
    uint16_t lquery[64];           // scoring side: the lower-cased term (search_field.rs:298-300); lm == 0xFFFFFFFF: the host scores the matches
};
struct DictMatch {  // one matched dictionary term, with what its score needs (search_field.rs:304-321, 691-732)
    uint32_t probe, term;
    uint32_t info;  // optimal-string-alignment distance of the lower-cased hit to the lower-cased term | plain Levenshtein distance << 8 (both
                    // capped at 255) | (lower-cased hit starts with the lower-cased term) << 16
};

struct UList {  // one input list of a union task (k_union)
    const uint32_t* docs;
    const void* scores;  // f16 anchor scores, or f32 values when flags & 1 (output of an earlier level)
    uint32_t len;
    float term_score;
    uint32_t flags, pad;
};
struct RangeJobD {  // k_range_hits: the leaf's lists [list_begin, + n_lists), its 1:n boost list's entry anchors [anchor_begin, + n_anchors) and the
                    // job's first block (n_lists == 1: 64 anchors per block, else one)
    uint32_t list_begin, n_lists, anchor_begin, n_anchors, block_begin, pad;
};
struct UTask {  // <= 64 lists merged by one wave per span
    uint32_t list_begin, n_lists;  // into the UList table
    uint32_t span_begin, n_spans;  // global span index of span 0; spans split the doc space at quantiles of list `pivot`
    uint32_t pivot, pad;           // absolute UList index of the longest list
};

struct FacetJob {
    uint32_t hist_off, num_values, top, out_off;
};

size_t tile_scan_lds_bytes(uint32_t n_bitmaps, uint32_t n_lists, uint32_t tile_words, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap, bool queue, uint32_t ml);

void launch_tile_scan(hipStream_t st, uint32_t total_spans, size_t lds_bytes, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                      const uint32_t* qmap, uint32_t nq, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist, bool queue, uint32_t ml,
                      bool facet_cache = false);
void launch_scan_leaf_f32(hipStream_t st, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base, const uint32_t* qmap,
                          uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist);
size_t scan_simple_lds_bytes(uint32_t cand_cap, uint32_t nv, uint32_t n_scatter, bool facet_cache);
uint32_t debug_div100_mismatches();
int debug_facet_select(const uint32_t* hist_host, uint32_t num_values, uint32_t top, uint32_t misalign, uint32_t* out_vals_host, uint32_t* out_counts_host);
void launch_scan_simple(hipStream_t st, bool wide, uint32_t n_scatter, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                        const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist, bool facet_cache = false);
size_t scan_probe_lds_bytes(uint32_t cand_cap, uint32_t nd);
void launch_scan_probe(hipStream_t st, uint32_t max_nd, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base, const uint32_t* qmap,
                       uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits, bool any_and = true, bool any_or = false);
// k_scan_ring (scan_ring.hip): persistent loader / consumer form of k_scan_probe.  `work`: two zeroed u32 (item counter, error word)
uint32_t scan_ring_consumers(uint32_t max_nd);
uint32_t scan_ring_slots(uint32_t maxnd, uint32_t consumers);
size_t scan_ring_lds_bytes(uint32_t maxnd, uint32_t consumers, uint32_t slots);
void launch_scan_ring(hipStream_t st, uint32_t max_nd, uint32_t grid, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* qmap, uint32_t nq, uint32_t spans_each,
                      const uint32_t* items, uint32_t total_items, uint32_t* work, unsigned long long* span_keys, unsigned long long* num_hits);
size_t scan_wide_lds_bytes(uint32_t cand_cap, uint32_t n_leaves, uint32_t n_scatter);
void launch_scan_wide(hipStream_t st, uint32_t max_leaves, uint32_t max_scatter, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                      const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits);
void launch_merge_spans(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const unsigned long long* span_keys,
                        unsigned long long* part_keys);
void launch_finalize(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const uint8_t* gathered, uint32_t num_shards,
                     size_t shard_stride, const PartialLayout& lay, uint32_t* res_ids, float* res_scores, uint32_t* res_n, unsigned long long* res_hits);
void launch_facet_select(hipStream_t st, uint32_t n_jobs, const FacetJob* jobs, const uint32_t* hist, uint32_t* out_vals, uint32_t* out_counts,
                         uint32_t* out_n);

void launch_range_hits(hipStream_t st, uint32_t n_blocks, uint32_t n_jobs, const UList* ulists, const RangeJobD* jobs, const uint32_t* anchors, unsigned long long* counts);
void launch_union(hipStream_t st, bool write, uint32_t total_spans, const UList* ulists, const UTask* tasks, const uint32_t* span_task, uint32_t* span_cnt,
                  const uint64_t* span_off, uint32_t* out_docs, float* out_vals, uint32_t* task_min);
void launch_scan_union(hipStream_t st, bool with_or, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base, const uint32_t* qmap,
                       uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits);
// all n_probes scan the SAME dictionary image (off / chars); matches are appended to out[0 .. out_cap) (the count keeps running beyond the cap)
void launch_dict_scan(hipStream_t st, const DictProbe* d_probes, uint32_t probe_base, uint32_t n_probes, const uint32_t* off, const uint16_t* chars, const uint16_t* low_chars,
                      uint32_t num_terms, uint32_t* out_count, uint32_t out_cap, DictMatch* out);

// ---- text locality pre-pass (K7)
struct LocRow {  // copy table[src .. src + len) to the gather buffer at dst
    uint64_t src, dst;
    uint32_t len, pad;
};
struct LocJob {  // one (request, field): its slice [seg_begin, seg_end) of the gathered text ids, its text_id_to_anchor rows, its output ranges
    const uint32_t* t2a_vals;
    const uint64_t* t2a_start;
    const uint32_t* t2a_len;
    uint32_t t2a_key_base, t2a_num_keys;
    uint32_t seg_begin, seg_end;
    uint32_t pair_begin, pair_end;  // slice of the (anchor, boost) pair buffer (known after the count pass)
    uint32_t out_off, pad;          // first entry of the job's result inside the output arrays
};
void launch_loc_gather(hipStream_t st, const LocRow* rows, uint32_t n_rows, const uint32_t* table, uint32_t* gathered);
void launch_loc_expand(hipStream_t st, bool write, const LocJob* jobs, uint32_t n_jobs, const uint32_t* sorted_text_ids, uint32_t n, uint32_t* totals_or_cursors,
                       unsigned long long* pairs);
void launch_loc_compact(hipStream_t st, const LocJob* jobs, uint32_t n_jobs, const unsigned long long* sorted_pairs, uint32_t* out_docs, float* out_vals, uint32_t* out_len);
size_t seg_sort_u32(void* tmp, size_t tmp_bytes, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t nseg, const uint32_t* seg_begin, const uint32_t* seg_end,
                    hipStream_t st);
size_t seg_sort_u64(void* tmp, size_t tmp_bytes, const unsigned long long* in, unsigned long long* out, uint32_t n, uint32_t nseg, const uint32_t* seg_begin,
                    const uint32_t* seg_end, hipStream_t st);

// ---- 1:n boost lists (K10, boost.rs:432-468)
struct B1nJob {  // sorted value ids [seg_begin, seg_end) -> the (anchor, boost value) pair of every boosted one, in value-id order
    const uint32_t* boost_present;  // bitmap over [boost_key_base, + boost_num_keys), or null = all present
    const float* boost_values;
    const uint64_t* to_anchor_off;  // value_id_to_anchor as a CSR
    const uint32_t* to_anchor_vals;
    uint32_t boost_key_base, boost_num_keys, to_anchor_key_base, to_anchor_num_keys;
    uint32_t seg_begin, seg_end, out_off, doc_lo, doc_hi, pad;
};
struct B1nResult {
    uint32_t len, total, flags, pad;  // pairs inside [doc_lo, doc_hi) / in all; flags: 1 anchors not ascending, 2 an anchor with several values
};
void launch_b1n_map(hipStream_t st, const B1nJob* jobs, uint32_t n_jobs, const uint32_t* sorted_value_ids, uint32_t* out_docs, float* out_vals, B1nResult* results);

// ---- explain: the returned hits' scores recomputed step by step, every intermediate value written to a trace
struct ExList {  // one posting list of one matched term (search_field.rs:419-444)
    const uint32_t* docs;
    const uint16_t* scores;
    uint32_t len;
    float term_score;
};
constexpr uint32_t XP_LEAF = 0, XP_AND = 1, XP_OR = 2;
struct ExOp {  // postfix program of the request's score tree (not limited by the scan kernels' descriptor sizes)
    uint32_t kind, nchild;
    uint32_t a;  // LEAF: first list | AND: offset of the summation order in aux[] | OR: offset of the operands' term slots in aux[]
    uint32_t b;  // LEAF: lists | OR: term slots
};
struct ExQuery {
    uint32_t op_begin, n_ops, list_begin, n_lists, col_begin, n_col;
    uint32_t doc_begin, trace_begin;  // first doc of the query in docs[]; first trace word of that doc
};
constexpr uint32_t kExStack = 256;  // operands alive at once
// trace of one doc, 3 words per entry: lists {f16 bits or 0xFFFFFFFF, anchor score, final score}, ops {present, value, OR: sum over the term slots},
// column boosts {applied, log10 factor, score after}, then {root present, tree score, final score}
__host__ __device__ inline uint32_t explain_trace_words(uint32_t n_lists, uint32_t n_ops, uint32_t n_col) { return 3u * (n_lists + n_ops + n_col + 1u); }
void launch_explain(hipStream_t st, uint32_t n_docs, const ExQuery* queries, const uint32_t* doc_query, const uint32_t* docs, const ExOp* ops, const uint16_t* aux,
                    const ExList* lists, const DColBoost* cols, uint32_t* trace);

#ifdef VQ_STAMP
void debug_read_stamps(unsigned long long* out, int reset);
void debug_read_probe_stamps(unsigned long long* out, int reset);
void debug_read_ring_stamps(unsigned long long* out, int reset);
#endif

}  // namespace vq
