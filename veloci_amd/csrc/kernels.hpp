// Host-callable launchers of kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_types.hpp"

namespace vq {

struct FacetJob {
    uint32_t hist_off, num_values, top, out_off;
};

size_t tile_scan_lds_bytes(uint32_t n_bitmaps, uint32_t n_lists, uint32_t tile_words, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap);

void launch_tile_scan(hipStream_t st, uint32_t total_spans, size_t lds_bytes, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                      const uint32_t* qmap, uint32_t nq, uint32_t stack_depth, uint32_t cand_cap, uint32_t desc_cap, unsigned long long* span_keys, unsigned long long* num_hits, uint32_t* hist);
size_t scan_simple_lds_bytes(uint32_t cand_cap, bool dense);
uint32_t debug_div100_mismatches();
void launch_scan_simple(hipStream_t st, bool dense, uint32_t total_spans, const uint8_t* blobs, const uint32_t* blob_off, const uint32_t* span_base,
                        const uint32_t* qmap, uint32_t nq, uint32_t cand_cap, unsigned long long* span_keys, unsigned long long* num_hits);
void launch_merge_spans(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const unsigned long long* span_keys,
                        unsigned long long* part_keys);
void launch_finalize(hipStream_t st, uint32_t nq, const uint8_t* blobs, const uint32_t* blob_off, const uint8_t* gathered, uint32_t num_shards,
                     const PartialLayout& lay, uint32_t* res_ids, float* res_scores, uint32_t* res_n, unsigned long long* res_hits);
void launch_hist_reduce(hipStream_t st, const uint8_t* gathered, uint32_t num_shards, const PartialLayout& lay, uint32_t* out);
void launch_facet_select(hipStream_t st, uint32_t n_jobs, const FacetJob* jobs, const uint32_t* hist, uint32_t* out_vals, uint32_t* out_counts,
                         uint32_t* out_n);

#ifdef VQ_STAMP
void debug_read_stamps(unsigned long long* out, int reset);
#endif

}  // namespace vq
