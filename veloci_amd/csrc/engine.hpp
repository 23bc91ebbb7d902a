// Host side of the MI355X query path: staged index image, query compiler, batch executor.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <map>
#include <limits>
#include <memory>
#include <thread>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "device_types.hpp"
#include "kernels.hpp"
#include "request.hpp"
#include "hostpool.hpp"

namespace vq {

using vqreq::VelociError;

#define VQ_HIP(expr)                                                                                                          \
    do {                                                                                                                      \
        hipError_t _e = (expr);                                                                                               \
        if (_e != hipSuccess) throw vqreq::VelociError(vqreq::ERR_DEVICE, std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr); \
    } while (0)

// suffix constants, reference src/persistence.rs:23-50
extern const char* const TOKENS_TO_TEXT_ID;
extern const char* const TO_ANCHOR_ID_SCORE;
extern const char* const PHRASE_PAIR_TO_ANCHOR;
extern const char* const VALUE_ID_TO_PARENT;
extern const char* const PARENT_TO_VALUE_ID;
extern const char* const TEXT_ID_TO_ANCHOR;
extern const char* const ANCHOR_TO_TEXT_ID;
extern const char* const BOOST_VALID_TO_VALUE;
extern const char* const VALUE_ID_TO_ANCHOR;
extern const char* const TEXTINDEX;
extern const char* const TOKEN_VALUES;

// ------------------------------------------------------------------ builder-side (host copies of what the caller hands over)
struct HostFst {
    std::vector<std::string> terms;  // bytewise sorted, ordinal == term id
};
struct HostPostings {
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> anchors;
    std::vector<uint32_t> scores;
    std::vector<uint64_t> global_lens;  // empty: lengths of the arrays handed over are the global ones
};
struct HostKV {
    uint32_t key_base = 0;
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> values;
};
struct HostPhrase {
    std::vector<uint32_t> t1, t2;
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> anchors;
};
struct HostBoost {
    uint32_t key_base = 0;
    std::vector<uint8_t> present;
    std::vector<uint32_t> bits;
};
struct ColumnMeta {
    bool is_anchor_identity_column = false;
    bool tokenize = true;
};

struct IndexBuilder {
    uint32_t num_anchors = 0, doc_lo = 0, doc_hi = 0;
    std::map<std::string, HostFst> fst;
    std::map<std::string, HostPostings> postings;
    std::map<std::string, HostKV> kv;
    std::map<std::string, HostPhrase> phrase;
    std::map<std::string, HostBoost> boost;
    std::map<std::string, ColumnMeta> columns;
};

// ------------------------------------------------------------------ staged image
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n);
    void ensure(size_t n);  // grow (never shrink); contents are not preserved
    void release();
    void upload(const void* src, size_t n, hipStream_t st = nullptr);
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct Dictionary {  // term dictionary of one text field
    std::vector<std::string> terms;
    std::unordered_map<std::string, std::vector<uint32_t>> lower_map;  // lowercase(term) -> ascending term ids (exact lookups)
    // device image for the fuzzy / prefix scan (k_dict_scan): code points as u16, raw and lower-cased
    bool bmp_only = true;  // false: some term has a code point above U+FFFF -> no device image, fuzzy unsupported on this field
    bool low_exact = true; // false: the lower-cased image is not str::to_lowercase of every term (U+0130): matches are scored on the host
    DevBuf d_off;          // u32 [T + 1]
    DevBuf d_raw;          // u16
    DevBuf d_low;          // u16
};

struct Index;
struct PostingStore;
struct FuzzyProbe {  // one dictionary scan of a batch (get_text_lines_from_fst, search_field.rs:68-99)
    std::string key;
    std::string path;                 // "<field>.textindex"
    std::vector<uint16_t> query;      // code points of the ORIGINAL term (lower-cased when case-insensitive)
    uint32_t max_d = 0;
    bool transposition = false, prefix = false, ci = true;
    std::vector<uint32_t> matches;    // ascending term ids == FST stream order
    std::vector<float> scores;        // default_score_for_distance of every match (search_field.rs:304-321), scored once per batch
    std::string lower_term;           // scoring side: the lower-cased term, the clamped distance, the prefix rule (:284-302)
    uint32_t lev = 0;
    bool check_prefix = false;
    int status = 0;
    std::string error;
};
using FuzzyTable = std::map<std::string, FuzzyProbe>;
std::string fuzzy_key(const vqreq::RequestSearchPart& p);
bool needs_dictionary_scan(const vqreq::RequestSearchPart& p);
void collect_fuzzy_probes(const Index& idx, const vqreq::Request& req, FuzzyTable& table);
struct Workspace;
void run_fuzzy_probes(const Index& idx, Workspace& ws, FuzzyTable& table, hipStream_t st);
void score_fuzzy_probe(const Index& idx, FuzzyProbe& probe);
float default_score_for_distance_host(uint8_t distance, bool prefix_matches);  // search_field.rs:27-33
size_t debug_sort_unique(uint32_t* ids, size_t n);
struct SuggestEntry {  // search_field.rs:158 SuggestFieldResult = Vec<(String, Score, TermId)>
    std::string text;
    float score;
    uint32_t term_id;
};
std::vector<SuggestEntry> suggest_part(const Index& idx, const vqreq::RequestSearchPart& part, const FuzzyTable* fuzzy);
void collect_suggest_probes(const Index& idx, const vqreq::Request& req, FuzzyTable& table);
std::optional<std::string> highlight_text(const std::string& text, const std::vector<std::string>& terms, const vqreq::SnippetInfo& opt, bool tokenized);  // highlight_field.rs:92-146
vqreq::Request page_request_after(const vqreq::Request& request, float score, uint32_t id);
std::vector<SuggestEntry> highlight_part(const Index& idx, const vqreq::RequestSearchPart& part, const FuzzyTable* fuzzy);
std::vector<SuggestEntry> run_highlight(const Index& idx, vqreq::RequestSearchPart part);  // search_field::highlight, search_field.rs:233-245
std::vector<SuggestEntry> run_suggest(const Index& idx, const vqreq::Request& req);  // suggest_multi, search_field.rs:194-219

// A leaf whose expansion matched many terms is materialised before the scan (k_union, K2): union of the
// terms' posting lists with the per-doc maximum of term_score * (f16 / 100).
struct UnionJob {
    std::string key;
    struct Term {
        const PostingStore* store;  // "<field>.textindex.to_anchor_id_score" of the term's field (a fused leaf spans several fields)
        uint32_t token;
        float score;                // term score
    };
    std::vector<Term> terms;                           // the lists to merge (those with entries in the unsharded index)
    // result (valid until the batch that ran it is finished; lives in the batch workspace)
    const uint32_t* d_docs = nullptr;
    const float* d_vals = nullptr;
    float max_value = std::numeric_limits<float>::infinity();  // largest value of the merged list (read back after the write pass)
    uint32_t len = 0;
    uint64_t global_len = 0;  // merged length over all shards
    uint64_t input_postings = 0;
};
using UnionTable = std::map<std::string, UnionJob>;
// Leaf hits inside doc ranges, counted before the final compilation (k_range_hits): the reference's merge of a 1:n boost list into a
// leaf's hits applies the first or all of an anchor's values depending on the hits around that anchor (boost.rs:255-281).
struct RangeJob {
    std::string key;
    std::string store_path;          // "<field>.textindex.to_anchor_id_score"
    std::vector<uint32_t> tokens;    // the leaf's posting lists
    std::string union_key;           // key of the leaf's union job when it asks to be materialised (the merged list is scanned instead)
    std::shared_ptr<const std::vector<uint32_t>> anchors;  // the boost list's entry anchors a_0 < a_1 < ... (owned by the batch's Boost1nEntry)
    std::vector<uint64_t> counts;    // result, summed over the shards: [2j] leaf postings at a_j, [2j + 1] strictly between a_(j-1) and a_j (0 for j = 0)
};
using RangeTable = std::map<std::string, RangeJob>;
// (anchor, boost value) lists of 1:n field boosts resolved on the host, shared by the requests and compilation passes of one batch
// One 1:n boost list of a batch (boost.rs:432-468 resolved for one leaf request and boost path), shared by the compilation passes and by every
// request of the batch with that leaf: the (anchor, value) pairs, what the compiler needs to know about their order, and — once known — the
// layers the device applies (one sorted unique list per applied value of an anchor).
struct Boost1nEntry {
    struct Layer {
        std::vector<uint32_t> docs;
        std::vector<float> vals;
        uint64_t global_len = 0;
    };
    std::once_flag resolved, layered;
    std::vector<std::pair<uint32_t, float>> pairs;
    bool ascending = true, several = false;
    std::vector<uint32_t> anchors;  // several: the distinct anchors, ascending
    std::shared_ptr<const std::vector<Layer>> layers;
};
struct Boost1nCache {  // sharded: every compiling thread comes here once per boosted leaf, and a thread that has to sleep on a lock loses more time
                       // than the lookup takes
    static constexpr size_t kShards = 64;
    struct Shard {
        std::mutex mu;
        std::unordered_map<std::string, std::shared_ptr<Boost1nEntry>> map;
    } shard[kShards];
    std::shared_ptr<Boost1nEntry> entry(const std::string& key) {
        Shard& sh = shard[std::hash<std::string>{}(key) % kShards];
        std::lock_guard<std::mutex> g(sh.mu);
        auto& slot = sh.map[key];
        if (!slot) slot = std::make_shared<Boost1nEntry>();
        return slot;
    }
    const std::map<std::string, struct Boost1nJob>* device = nullptr;  // lists the K10 pre-pass has resolved on the device (second compilation pass on)
};
// Text locality of a field whose text ids are not anchors (boost.rs:34-87), resolved before the final compilation by the K7 pre-pass: the
// token -> text rows of the query's terms are gathered and sorted, texts occurring c > 1 times are expanded to their anchors with the boost
// 2*c*c, the (anchor, boost) pairs are sorted and the smallest boost of every anchor is kept (the reference's reversed max_by, boost.rs:25).
struct LocalityJob {
    std::string key;
    std::string t2t_path, t2a_path;                   // "<field>.textindex.tokens_to_text_id" / ".text_id_to_anchor"
    std::vector<uint32_t> tokens;                     // token ids of all terms (multiplicity kept: the count is over list entries, boost.rs:51-56)
    // result (valid until the batch that ran it is finished; lives in the batch workspace)
    const uint32_t* d_docs = nullptr;
    const float* d_vals = nullptr;
    uint32_t len = 0;
};
using LocalityTable = std::map<std::string, LocalityJob>;
// A 1:n boost list (boost.rs:432-468) resolved by the K10 pre-pass: the leaf's text ids -> value ids of the 1:n object (gathered and sorted on
// the device) -> the (anchor, boost value) pair of every boosted value id, in value-id order, as a (doc, f32) list the scans read.
struct Boost1nJob {
    std::string key;
    std::string to_parent_path, to_anchor_path, boost_path;  // "<leaf>.textindex.value_id_to_parent", "<boost>.value_id_to_anchor", "<boost>.boost_valid_to_value"
    std::vector<uint32_t> text_ids;
    // result (lives in the batch workspace)
    const uint32_t* d_docs = nullptr;
    const float* d_vals = nullptr;
    uint32_t len = 0;          // pairs inside this shard
    uint32_t total = 0;        // pairs of the unsharded list
    bool ascending = true;     // anchors non-decreasing in value-id order (else the reference's merge is not reproducible: declined)
    bool several = false;      // some anchor has more than one boosted value: the look-ahead rule applies (host path)
    bool done = false;
};
using Boost1nTable = std::map<std::string, Boost1nJob>;
constexpr int kStatusNeedsUnion = -1;  // internal: compile again once the requested union / locality jobs have run
constexpr int kStatusNeedsCounts = -2; // internal: the compiled query IS a count pre-pass; compile again with its results
constexpr int kStatusNeedsRanges = -3; // internal: compile again once the requested range jobs have run

// Result sizes of the operands of AND nodes, measured by a count pre-pass (the reference orders the score sum of an AND by
// its operands' result lengths and labels an AND result by them, set_op.rs:388-393,439).
struct QueryCounts {
    bool has_filter = false;
    uint64_t filter_count = 0;                                       // ids in the filter result (Set below 100 001 ids, filter_result.rs:5-22)
    std::map<uint32_t, std::pair<uint64_t, uint64_t>> nodes;        // node id (preorder) -> (hits, hits inside the filter)
};

struct PostingStore {  // "<field>.textindex.to_anchor_id_score": padded segmented arrays in HBM
    uint32_t num_tokens = 0;
    std::vector<uint64_t> start;        // start[t]: first entry of list t inside docs/scores (multiple of 4)
    std::vector<uint32_t> len;          // entries of list t inside this shard
    std::vector<uint64_t> global_len;   // entries of list t in the unsharded index
    std::vector<uint16_t> max_raw;      // largest f16 score bits of list t (upper bound for OR pruning)
    DevBuf docs;                        // u32, lists padded to a multiple of 4 with 0xFFFFFFFF
    DevBuf scores;                      // f16 bits (u16), same indexing
    uint64_t total_padded = 0;
    // dense lists (>= 1/64 of the shard's docs) also get a bitmap image + a rank directory
    std::vector<int64_t> bm_start;      // word offset of list t inside `bitmaps`, or -1
    std::vector<int64_t> rd_start;      // entry offset of list t inside `rank_dir`, or -1
    DevBuf bitmaps;                     // u32 words; bit (doc - Index::bitmap_base)
    DevBuf rank_dir;                    // u32: entries below bitmap_base + (k << kRankShift)
    // lists that hold at least 1/4096 of the shard's docs get a tile directory (k_scan_probe reads a tile's postings of
    // its cover list without searching or counting): entries below bitmap_base + (k << kTileDirShift)
    std::vector<int64_t> td_start;      // entry offset of list t inside `tile_dir`, or -1
    DevBuf tile_dir;
    // ... and a tile-packed image for k_scan_probe (round 4): the list's postings grouped by 32768-doc tile, every tile padded to a
    // multiple of 8 entries (a "granule"), as `cov32` = (doc - tile_lo) << 16 | f16 score (what a COVER streams: 4 B per posting, one
    // 16 B/lane load per 256 postings, no index arithmetic) and — when no tile holds more than 2048 entries — as `arr16` = doc - tile_lo
    // (what an OPERAND of fewer than 1/16 of the docs is probed in: a sorted 16-bit array per tile, Roaring's array container, 2 B per
    // posting instead of a bit per doc).  Both share one directory: gdir[k] = granules below tile k.
    std::vector<int64_t> pk_start;      // granule offset of list t inside `cov32`, or -1
    std::vector<int64_t> ak_start;      // granule offset of list t inside `arr16`, or -1 (a tile with more than 2048 entries)
    std::vector<int64_t> gd_start;      // entry offset of list t inside `gdir`
    DevBuf cov32, arr16, gdir;
};

struct KVStore {  // IndexIdToParent<u32>: host copy + (where useful) device images
    uint32_t key_base = 0, num_keys = 0;  // host addressing, as handed over
    std::vector<uint64_t> host_off;
    std::vector<uint32_t> host_values;
    // list use: values are anchors -> every row is a padded, sorted, unique doc-id list in HBM
    bool list_rows = false;
    bool rows_sorted_unique = true;
    bool rows_equal_postings = false;  // tokens_to_text_id of an identity column: row t holds exactly the docs of posting list t
    std::vector<uint64_t> start;  // first entry of row r inside `values` (multiple of 4)
    std::vector<uint32_t> len;    // entries of row r inside this shard
    DevBuf values;
    DevBuf d_row_start, d_row_len;  // text_id_to_anchor: `start` / `len` on the device (u64 / u32 per row) — text locality expands text ids there (K7)
    // text use: tokens_to_text_id of a field whose text ids are NOT anchors: the whole table as a CSR in HBM (values are text ids: replicated
    // on every shard like the dictionary), read by the text-locality pre-pass (K7)
    bool text_csr = false;
    DevBuf d_text_vals;  // u32, host_values as handed over (row r = [host_off[r], host_off[r + 1]))
    // 1:n boost use (K10): value_id_to_parent (text id -> value ids) and value_id_to_anchor (value id -> anchor) as whole CSRs in HBM: keys
    // are not anchors, so the tables are replicated on every shard like the dictionary; d_text_vals holds the values, d_csr_off the offsets
    bool value_csr = false;
    DevBuf d_csr_off;    // u64 [num_keys + 1]
    // facet use: keys are anchors -> CSR restricted to the shard's anchors
    bool facet_csr = false;
    uint32_t csr_key_base = 0, csr_num_keys = 0;
    uint32_t csr_max_value = 0;  // largest value id of the CSR (text ids of texts too long for the dictionary lie beyond it)
    DevBuf csr_direct;  // u32 [csr_num_keys], only when no anchor has more than one value: the value id or 0xFFFFFFFF (one gather per hit
                        // instead of two offsets + the value)
    DevBuf csr_off;     // u64 [csr_num_keys + 1]
    DevBuf csr_values;  // u32

    bool host_row(uint64_t key, const uint32_t** b, const uint32_t** e) const {  // IndexIdToParent::get_values
        if (key < key_base) return false;
        uint64_t r = key - key_base;
        if (r >= num_keys || host_off[r] == host_off[r + 1]) return false;
        *b = host_values.data() + host_off[r];
        *e = host_values.data() + host_off[r + 1];
        return true;
    }
};

struct PhraseStore {
    std::vector<std::pair<uint32_t, uint32_t>> keys;  // sorted
    std::vector<uint64_t> start;
    std::vector<uint32_t> len;
    DevBuf anchors;  // padded rows
};

struct BoostColumn {
    uint32_t key_base = 0, num_keys = 0;
    bool has_present = false;
    float vmin = 0.0f, vmax = 0.0f;  // smallest / largest present value (any_value false: none)
    bool any_value = false, has_nan = false;
    DevBuf values;   // f32
    DevBuf present;  // bitmap u32
    std::vector<uint32_t> host_bits;    // host copy, kept for 1:n boost columns (keys are value ids, resolved on the host)
    std::vector<uint8_t> host_present;
    bool host_value(uint64_t key, float* out) const {  // IndexIdToParent::get_value
        if (key < key_base || key - key_base >= host_bits.size()) return false;
        const size_t r = size_t(key - key_base);
        if (!host_present.empty() && !host_present[r]) return false;
        std::memcpy(out, &host_bits[r], 4);
        return true;
    }
};

struct PinnedBuf {  // page-locked host staging
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf();
    void ensure(size_t n);
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr int kWorkspaces = 4;  // batches in flight per index (host compile of one overlaps the scan of the others; the chunks of one sharded step: one each)

// Kernels the profiler accounts separately (vq_profile_json): the pre-passes, one entry per scan class, the merges.
enum KernelId : int {
    K_DICT_SCAN = 0, K_UNION_COUNT, K_UNION_WRITE, K_RANGE_HITS, K_COUNT_PREPASS, K_SCAN_LEAF_F32, K_SCAN_RICH, K_SCAN_RING, K_SCAN_PROBE, K_SCAN_AND, K_SCAN_SIMPLE, K_SCAN_UNION,
    K_SCAN_WIDE, K_TILE_SCAN, K_MERGE_SPANS, K_FINALIZE, K_FACET_SELECT, K_LOCALITY, K_BOOST1N, K_COUNT_
};
extern const char* const kKernelNames[K_COUNT_];

struct TimedLaunch {  // one profiled launch of a batch: events [begin, end] on the launch stream
    int kernel;
    uint32_t ev_begin, ev_end;
    uint64_t layout_bytes, algorithmic_bytes, queries;
};

struct Workspace {  // scratch of one in-flight batch
    std::mutex mu;
    std::atomic<bool> pinned{false};  // held by a batch that named this workspace (the chunks of a sharded step): callers that take any free one skip it
    std::vector<hipEvent_t> ev_pool;          // profiling: created on first use
    uint32_t ev_used = 0;
    std::vector<TimedLaunch> timed;           // launches of the batch in flight (profiling)
    hipEvent_t ev_done = nullptr;             // scan + span merge finished: the finish stream waits on it
    PinnedBuf h_up, h_down;
    DevBuf d_up;        // blobs + blob_off + span_base + facet jobs
    DevBuf d_span_keys;
    DevBuf d_partial;
    DevBuf d_down;      // results
    DevBuf d_union_docs[2], d_union_vals[2], d_union_max, d_union_meta;  // materialised leaves (k_union), level 1 / level 2
    DevBuf d_loc_a, d_loc_b, d_loc_pairs_a, d_loc_pairs_b, d_loc_meta, d_loc_tmp, d_loc_docs, d_loc_vals;  // text locality pre-pass (K7)
    DevBuf d_b1n_a, d_b1n_b, d_b1n_meta, d_b1n_tmp, d_b1n_docs, d_b1n_vals;                              // 1:n boost lists (K10)
    DevBuf d_probe_desc, d_probe_counts, d_probe_ids;                    // dictionary scans (k_dict_scan): kept, so that no hipFree synchronises the device mid-pipeline
};

struct KernelProfile {
    double ms = 0;
    uint64_t launches = 0;
    uint64_t layout_bytes = 0;       // bytes THIS layout has to move for the launches (bitmap words of dense lists, 4 B per id of scattered lists,
                                     // 6 B per streamed posting, gathered bytes counted by the kernels, 8 B per returned key): roofline numerator
    uint64_t algorithmic_bytes = 0;  // SURVEY.md 8(d) accounting (6 B per posting of every list ...): what a posting-streaming design would move
    uint64_t gathered_bytes = 0;     // the part of layout_bytes read by per-hit gathers (2 B per f16 score, 4 B per f32 / column / facet value, ...)
    uint64_t queries = 0;
};
struct Profile {
    bool enabled = false;
    KernelProfile k[K_COUNT_];
    uint64_t batches = 0;
};
// RAII bracket of one launch (or a few back-to-back launches of one kernel) with two events of the workspace's pool
struct LaunchTimer {
    Workspace* ws = nullptr;
    hipStream_t st = nullptr;
    size_t slot = 0;
    LaunchTimer(bool on, Workspace& w, hipStream_t s, int kernel, uint64_t layout_bytes = 0, uint64_t algorithmic_bytes = 0, uint64_t queries = 0);
    ~LaunchTimer();
    LaunchTimer(const LaunchTimer&) = delete;
    LaunchTimer& operator=(const LaunchTimer&) = delete;
};

// A few persistent host threads for the per-request work of a batch (query compilation): spawning threads per chunk costs more
// than compiling a small chunk.

// The exchange step of a sharded search inside the library (vq_comm_*): RCCL — loaded at run time, one communicator per index / rank — or the
// caller's own all-gather / all-reduce (vq_comm_init_custom: rehearsals with several shards in one process).  Collectives run on their own
// stream between the scan stream and the finish stream, so that step i's exchange and merge overlap step i+1's scans (two steps in flight).
struct ShardComm {
    int nranks = 1, rank = 0;
    void* nccl = nullptr;  // ncclComm_t
    int (*allgather)(void* ctx, const void* local, void* gathered, size_t bytes_per_rank, void* stream) = nullptr;
    int (*allreduce_u32)(void* ctx, void* inout, size_t count, void* stream) = nullptr;
    void* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_scan[2] = {nullptr, nullptr}, ev_xchg[2] = {nullptr, nullptr}, ev_fin[2] = {nullptr, nullptr};
    void* unmerged = nullptr;  // the step in flight whose merge is not queued yet (vq_shard_step*)
    int live = 0;              // steps begun and not yet ended / freed (at most two: each holds workspaces of its own)
    DevBuf gathered[2];
    DevBuf red;  // scratch of the sums-over-shards hook (u64)
    bool busy[2] = {false, false};  // parity (workspace pair, gather buffer, events) held by a live step
    std::string failed;  // != "": a step of this communicator failed as a whole on this rank (or did not complete in time): the ranks' exchanges no longer line
                         // up, every further step is refused until the communicator is made again (vq_comm_init / vq_comm_init_custom)
    std::mutex mu;  // the hook may be called from compile threads of two steps
    ~ShardComm();
};

struct Index {
    int device = 0;
    // sums over all shards (vq_index_set_allreduce); null: requests that need them are declined on a sharded index
    int (*allreduce_fn)(void*, uint64_t*, size_t) = nullptr;
    void* allreduce_ctx = nullptr;
    bool sharded() const { return doc_lo != 0 || doc_hi != num_anchors; }
    bool can_sum_over_shards() const { return !sharded() || allreduce_fn != nullptr; }
    void sum_over_shards(std::vector<uint64_t>& v) const {
        if (!sharded() || v.empty()) return;
        if (!allreduce_fn || allreduce_fn(allreduce_ctx, v.data(), v.size()) != 0) throw vqreq::VelociError(vqreq::ERR_DEVICE, "all-reduce over the shards failed");
    }
    mutable std::atomic<uint64_t> or_reruns{0};  // requests that ran a second time because k_scan_probe_or's short cut could not be confirmed (tests, tools)
    mutable std::unique_ptr<HostPool> pool;  // created on first use
    mutable std::mutex pool_mu;
    uint32_t num_anchors = 0, doc_lo = 0, doc_hi = 0;
    uint32_t bitmap_base = 0;   // doc id of bit 0 of the list bitmaps: doc_lo rounded down to 65536
    uint64_t bitmap_words = 0;  // words of one list bitmap
    std::map<std::string, Dictionary> dict;
    std::map<std::string, PostingStore> postings;
    std::map<std::string, KVStore> kv;
    std::map<std::string, PhraseStore> phrase;
    std::map<std::string, BoostColumn> boost;
    std::map<std::string, ColumnMeta> columns;
    uint64_t device_bytes = 0;
    // facets on 1:n fields without an anchor_to_text_id index (facet.rs:59-70): the chain of parent_to_value_id joins is composed
    // once per field, on first use, into one anchor-keyed CSR that the facet kernel reads like a direct index
    mutable std::mutex composed_mu;
    mutable std::map<std::string, std::unique_ptr<KVStore>> composed_facets;
    const KVStore& composed_facet(const std::vector<std::string>& steps) const;
    hipStream_t own_stream = nullptr, own_fin_stream = nullptr;
    hipStream_t pre_stream = nullptr;  // dictionary scans, union / range / count pre-passes: every one ends in a host synchronisation, so they need no
                                       // ordering with the scan stream — and on their own stream the pre-passes of batch c+1 overlap the scan of batch c
    hipStream_t stream = nullptr;      // scans + span merges
    hipStream_t fin_stream = nullptr;  // shard merge, facet selection, result download (== stream when the caller set one)
    mutable std::mutex profile_mutex;
    mutable Profile profile;
    mutable Workspace ws[kWorkspaces];
    mutable std::atomic<uint32_t> next_ws{0};
    // Sharded step run as a pipeline of chunks with ONE collective: the chunks' partials are placed back to back in this arena (run_partial's
    // arena_offset), so a single all-gather of its used prefix exchanges them all; every chunk then merges out of the gathered copy with the
    // prefix's size as the shard stride.  Fixed size: a step that does not fit is told so and takes the per-chunk path.
    mutable DevBuf arena;
    static constexpr size_t kArenaBytes = 64ull << 20;
    mutable std::unique_ptr<ShardComm> comm;  // vq_comm_init / vq_comm_init_custom
    ~Index();
    bool is_anchor_identity(const std::string& textindex_path) const;
};

std::unique_ptr<Index> build_index(const IndexBuilder& b, int device);

// ------------------------------------------------------------------ compiled query
struct HList {
    const uint32_t* d_docs = nullptr;
    const uint16_t* d_scores = nullptr;
    uint32_t len = 0;
    uint32_t flags = 0;
    float term_score = 0.f;
    uint16_t max_raw = 0x7C00;  // +inf: no bound known
    float max_value = std::numeric_limits<float>::infinity();  // LIST_F32 (materialised leaf): its largest value
    uint64_t global_len = 0;
    const uint32_t* d_bitmap = nullptr;
    const uint32_t* d_rank_dir = nullptr;
    const uint32_t* d_tile_dir = nullptr;
    const uint32_t* d_cov32 = nullptr;  // tile-packed image (PostingStore::cov32 / arr16 / gdir), null without one
    const uint16_t* d_arr16 = nullptr;
    const uint32_t* d_gdir = nullptr;
    int inline_idx = -1;  // >= 0: docs come from inline_lists[inline_idx] (carried inside the blob)
    int inline_val_idx = -1;  // >= 0: f32 values come from inline_vals[inline_val_idx]
};

struct FacetOut {
    std::string field;
    std::string dict_path;  // fst used to turn value ids into strings
    uint32_t top = 10;
    uint32_t num_values = 0;
    uint32_t host_top = 0;  // > 0: more entries than k_facet_select ranks (kMaxTopK) — the job's histogram is downloaded and ranked on the host; `top` is 0 then
};

// ---- explain (src/search/result/explain.rs:2-21)
struct ExplainRec {
    enum Kind : uint8_t { Boost, MaxTokenToTextId, TermToAnchor, LevenshteinScore, OrSumOverDistinctTerms } kind = Boost;
    float a = 0.0f, b = 0.0f, c = 0.0f;  // Boost(a) | TermToAnchor{term_score a, anchor_score b, final_score c} | LevenshteinScore{score a} | OrSum(a)
    uint32_t term_id = 0;
    std::string text;  // LevenshteinScore.text_or_token_id
};
using ExplainRecs = std::vector<ExplainRec>;
struct ExplainNode {  // one node of the request's score tree, as the reference executes it (no leaf fusion, no materialised unions)
    uint32_t kind = 0;               // XP_LEAF / XP_AND / XP_OR
    int op = -1;                     // its op in ExplainPlan::ops
    std::vector<int> children;       // request order
    std::vector<uint16_t> order;     // AND: the operands in summation order (set_op.rs:393), the shortest one last
    // leaf: one list per matched term, in the order of the dictionary result's hits (search_field.rs:419)
    uint32_t list_begin = 0, list_count = 0;
    std::vector<uint32_t> list_term;
    std::map<uint32_t, ExplainRecs> term_records;  // the dictionary result's explain map, keyed by TERM id (search_field.rs:336-343; field_result.rs:44 copies it
                                                   // into the anchor-keyed result: an anchor whose id is one of these keys starts with that term's records)
};
using WhyFoundPlan = std::map<std::string, std::vector<uint32_t>>;
struct ExplainPlan {
    std::vector<ExplainNode> nodes;
    int root = -1;
    std::vector<ExList> lists;
    std::vector<ExOp> ops;
    std::vector<uint16_t> aux;
    std::vector<DColBoost> cols;  // the request-level column boosts (boost.rs:470-504), request order
};

struct CompiledQuery {
    size_t blob_bytes = 0, desc_bytes = 0;  // size of the packed blob / of its descriptor part, taken right after compilation on the compiling thread (0: not taken)
    int status = 0;
    std::shared_ptr<const ExplainPlan> explain_plan;  // request.explain: what complete_explain_requests needs for the returned hits
    // request.why_found with request.select (search.rs:220-224): textindex path -> all term ids the search matched there (what get_why_found flattens out of
    // term_id_hits_in_field, why_found.rs:24-27); complete_why_found_requests highlights the returned anchors' texts with them
    std::shared_ptr<const WhyFoundPlan> why_found_plan;
    std::string error;
    std::vector<RangeJob> range_requests;  // status == kStatusNeedsRanges
    std::vector<UnionJob> union_requests;  // status == kStatusNeedsUnion: jobs to run before compiling again
    std::vector<LocalityJob> locality_requests;  // likewise (K7)
    std::vector<Boost1nJob> boost1n_requests;    // likewise (K10)
    std::vector<uint32_t> count_nodes;     // status == kStatusNeedsCounts: node ids; counters 2i / 2i+1 = hits / hits inside the filter, then the filter
    uint32_t n_counts = 0;
    std::vector<HList> lists;
    std::vector<std::vector<uint32_t>> inline_lists;
    std::vector<std::vector<float>> inline_vals;
    std::vector<DColBoost> leaf_cols;  // compile-time staging of the OP_BOOST1N parameters
    uint32_t seq_tiles = 0;          // see QHeader::seq_tiles
    uint32_t prune_n = 0;            // see QHeader::prune_n
    uint64_t prune_mask = 0;
    uint32_t prune_gbits[16] = {};
    uint32_t n_top_cols = 0;  // cols[0 .. n_top_cols) are the request-level boosts; the rest belong to OP_BOOST1N ops
    std::vector<DOp> ops, fops;
    std::vector<DPresOp> pres;
    std::vector<uint16_t> pres_in;
    uint32_t n_temps = 0;
    uint32_t simple_n = 0;
    uint32_t simple_flags = 0;
    std::vector<DGroup> groups;
    std::vector<DTermBoost> tboosts;
    std::vector<DColBoost> cols;
    std::vector<DLocField> locf;
    std::vector<uint16_t> loc_idx;
    DSimple2 simple2{};  // simple_flags bit 18
    DWide wide{};        // simple_flags bit 24
    DProbe probe{};      // simple_flags bit 25
    std::vector<DFacet> facets;
    std::vector<FacetOut> facet_out;
    std::map<std::string, std::vector<std::string>> why_found_terms;  // search.rs:186: path -> matched term texts (request.why_found)
    uint32_t top = 10, skip = 0, top_k = 10;
    uint64_t total_len = 0;       // sum of shard-local list lengths (work estimate)
    uint64_t algorithmic_bytes = 0;
    uint64_t layout_bytes = 0;   // static part of KernelProfile::layout_bytes for this query (set when the kernel route is known)
    uint64_t key_upper = ~0ull;  // QHeader::key_upper
    // k_scan_probe_or (simple_flags bit 27) scores only the docs that hold the cover; a doc without it scores at most this (the OR formula on the
    // other operands' list maxima).  The finished request is exact if its k-th best key lies above: finish_batch checks and otherwise asks for a
    // second run without the short cut (Result::rerun_exact).  NaN: not such a query.
    float or_skip_bound = std::numeric_limits<float>::quiet_NaN();
    bool deep = false;           // top + skip > kMaxTopK: this compilation ranks the first kMaxTopK only; the caller pages on (search_pages)
    uint32_t tile_words = 0, n_spans = 1, stack_depth = 1;
    uint32_t max_spans = 1;  // tiles of the query's doc range (<= 4096): a small batch splits its queries further, up to this (exec.cpp)
};

CompiledQuery compile_query(const Index& idx, const vqreq::Request& req, const FuzzyTable* fuzzy = nullptr, const UnionTable* unions = nullptr,
                            const QueryCounts* counts = nullptr, const RangeTable* ranges = nullptr, Boost1nCache* boost_cache = nullptr,
                            const LocalityTable* localities = nullptr);
void run_locality_jobs(const Index& idx, Workspace& ws, LocalityTable& table, hipStream_t st);
void run_boost1n_jobs(const Index& idx, Workspace& ws, Boost1nTable& table, hipStream_t st);
void run_range_jobs(const Index& idx, Workspace& ws, RangeTable& table, const UnionTable& unions, hipStream_t st);
void run_union_jobs(const Index& idx, Workspace& ws, UnionTable& table, hipStream_t st);
struct Result;
// Deep requests (top + skip > kMaxTopK) of an unsharded batch: results[i] holds page 0; fetch the following pages (each one scan that
// ranks only keys below the previous page's last) and cut the requested window.
// explain records of the returned hits (k_explain + host formatting); requests without `explain` are left alone
void complete_explain_requests(const Index& idx, std::vector<std::unique_ptr<Result>>& results, std::vector<int>& status, std::vector<std::string>& errors);
// why_found_info of the returned hits (get_why_found, search/why_found.rs:11-50): host work over the host copies of the key-value stores and the
// dictionary, on the final window of ids; requests without `why_found` + `select` are left alone
void complete_why_found_requests(const Index& idx, std::vector<std::unique_ptr<Result>>& results, std::vector<int>& status, std::vector<std::string>& errors);
bool request_wants_explain(const vqreq::Request& req);
std::string explain_records_json(const ExplainRecs& records);
void complete_deep_requests(const Index& idx, const vqreq::Request* const* reqs, size_t n, std::vector<std::unique_ptr<Result>>& results,
                            std::vector<int>& status, std::vector<std::string>& errors);

// ------------------------------------------------------------------ results
struct ResultFacet {
    std::string field;
    std::vector<std::pair<std::string, uint64_t>> entries;
};
struct Result {
    uint64_t num_hits = 0;
    uint64_t execution_time_ns = 0;
    std::vector<uint32_t> ids;
    std::vector<float> scores;
    std::vector<ResultFacet> facets;
    bool has_facets = false;
    std::map<std::string, std::vector<std::string>> why_found_terms;
    std::shared_ptr<const ExplainPlan> explain_plan;
    std::shared_ptr<const WhyFoundPlan> why_found_plan;
    std::map<uint32_t, std::map<std::string, std::vector<std::string>>> why_found_info;  // search.rs:220-224: anchor -> field -> highlighted texts
    bool has_explain = false;
    std::vector<std::pair<bool, ExplainRecs>> explain;  // per returned hit: (the reference's map has an entry for the hit, its records) — search.rs:86,96
    bool deep = false;  // holds the first page of a deep request (see CompiledQuery::deep)
    bool rerun_exact = false;  // k_scan_probe_or's short cut could not be confirmed (CompiledQuery::or_skip_bound): finish_batch runs the request again
    mutable std::string json;
};

struct PartialBatch {
    const Index* index = nullptr;
    Workspace* ws = nullptr;
    std::unique_lock<std::mutex> lock;    // holds the workspace until the batch is finished
    bool pinned_ws = false;               // ... a workspace the caller named (Workspace::pinned is ours to clear)
    std::vector<CompiledQuery> queries;   // status != 0: failed at compile time
    std::vector<uint32_t> slot;           // slot[i]: position of request i among the device queries, or UINT32_MAX
    std::vector<uint8_t> qclass;          // profiling: KernelId of the scan that serves device query q
    std::vector<FacetJob> facet_jobs;     // host copy of the facet jobs (histogram offsets of the ones ranked on the host)
    uint32_t nq_dev = 0;
    PartialLayout layout{};
    // device addresses inside the workspace
    const uint8_t* d_blobs = nullptr;
    const uint32_t* d_blob_off = nullptr;
    const uint32_t* d_span_base = nullptr;
    const FacetJob* d_facet_jobs = nullptr;
    uint8_t* d_partial = nullptr;
    uint32_t total_spans = 0;
    uint32_t n_facet_jobs = 0;
    uint32_t total_facet_out = 0;
    bool profiled = false;
    bool launched = false, finished = false;  // scans queued / results taken
    bool merge_launched = false;              // finish_batch phase 1 done (merge + download queued on the finish stream)
    std::chrono::steady_clock::time_point t0;
    void release_workspace();
    std::vector<const vqreq::Request*> reqs;  // the batch's requests as handed in (alive until the batch is finished: a request may have to run again)
    ~PartialBatch();  // a batch given up before its merge waits for its scans: the workspace (pinned staging, blobs) is handed on only when the device is done with it
};

std::unique_ptr<PartialBatch> run_partial(const Index& idx, const vqreq::Request* const* reqs, size_t n, int slot = -1, int64_t arena_offset = -1);
void finish_batch(const Index& idx, PartialBatch& pb, const void* gathered_device, uint32_t num_shards, std::vector<std::unique_ptr<Result>>& out,
                  std::vector<int>& status, std::vector<std::string>& errors, size_t shard_stride = 0);  // shard_stride: bytes between the shards'
                                                                                                         // copies in `gathered_device` (0: the partial's own size)
// the two halves of finish_batch: queue merge + download on the finish stream (nothing is waited for) / wait for them and build the results
void finish_launch(const Index& idx, PartialBatch& pb, const void* gathered_device, uint32_t num_shards, size_t shard_stride = 0);

}  // namespace vq
