/*
 * veloci_amd.h — C ABI of the MI355X-native veloci query-execution path.
 *
 * This is the drop-in boundary.  The reference exposes the path as
 *
 *     pub fn search(request: Request, persistence: &Persistence)
 *         -> Result<SearchResult, VelociError>              (src/search.rs:143)
 *
 * and has no FFI of its own; every entry point below states which reference
 * interface it replaces.  Plain pointers and sizes only: a Rust `extern "C"`
 * block, cgo or ctypes can bind it (INTEGRATION.md shows the Rust stub).
 *
 * Threading: a `vq_index` is immutable after `vq_index_build`; `vq_search*`
 * may be called from many host threads on one index (reference: `Persistence:
 * Sync`, src/persistence.rs:80-84).  Errors: every call returns `VQ_OK` or an
 * error code and leaves a message for `vq_last_error()` (thread local) that
 * reproduces the reference's `VelociError` rendering (src/error.rs:5-43).
 */
#ifndef VELOCI_AMD_H
#define VELOCI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- errors */

enum {
    VQ_OK = 0,
    /* VelociError::InvalidRequest (src/error.rs:11) and the reference's panics on
       malformed plans (execution_plan.rs:220,408,437; plan_steps.rs:287) */
    VQ_ERR_INVALID_REQUEST = 1,
    /* VelociError::FstNotFound: "field does not exist {path} (fst not found)" (src/error.rs:35) */
    VQ_ERR_FST_NOT_FOUND = 2,
    /* VelociError::StringError("Did not found path in indices ...") (src/persistence.rs:454-458) */
    VQ_ERR_INDEX_NOT_FOUND = 3,
    /* request uses a feature outside the GPU hot path (the declined explain combinations, why_found_info on the flat / sharded entry points, the bounds of DESIGN.md §7 ...): never silently ignored */
    VQ_ERR_UNSUPPORTED = 4,
    /* HIP runtime failure / no device / extension missing */
    VQ_ERR_DEVICE = 5,
    VQ_ERR_INVALID_ARGUMENT = 6,
    /* serde_json parse failure of the request text (VelociError::JsonError) */
    VQ_ERR_JSON = 7
};

/* Message of the last failing call on this thread ("" if none). */
const char* vq_last_error(void);

/* ------------------------------------------------------- index (load side)
 *
 * Replaces `Persistence::load` + `PersistenceIndices` (src/persistence.rs:52-60,
 * 206-291, 393-410) at the *decoded array* level: the caller (a Rust shim walking
 * `persistence.indices`, or the synthetic generator) hands over every index as
 * plain CSR arrays under the reference's own index names
 * ("<field>.textindex.to_anchor_id_score", "...tokens_to_text_id", ...,
 * suffix constants src/persistence.rs:23-50).  Arrays are copied; the caller may
 * free them after the call returns.
 *
 * Sharding: an index holds the anchors (documents) in
 * [doc_lo, doc_hi).  Anchor-valued lists may be passed whole or pre-sliced —
 * the loader drops anchors outside the range.  Anchor-keyed stores take
 * `key_base`: the id of the first key in the arrays handed over.
 */
typedef struct vq_index_builder vq_index_builder;
typedef struct vq_index vq_index;

vq_index_builder* vq_index_builder_new(uint32_t num_anchors, uint32_t doc_lo, uint32_t doc_hi);
void vq_index_builder_free(vq_index_builder*);

/* Term dictionary of "<field>.textindex" — replaces `indices.fst[path]`
 * (src/persistence.rs:59; built at src/create/create_fulltext.rs:53-80).
 * Terms are UTF-8, bytewise sorted, ordinal == term id. */
int vq_index_add_fst(vq_index_builder*, const char* path, uint32_t num_terms,
                     const uint8_t* term_bytes, const uint64_t* term_offsets /* [num_terms+1] */);

/* Posting lists "<field>.textindex.to_anchor_id_score" — replaces
 * `TokenToAnchorScore::get_score_iter` (src/persistence.rs:80-82,
 * src/indices/persistence_score/token_to_anchor_score_vint.rs:150-204).
 * Per token: anchors ascending & unique, integer scores as stored
 * (converted to f16 with RNE at load, as `f16::from_f32(score as f32)` :155).
 * `global_lens` (may be NULL) gives each list's length in the *unsharded* index;
 * needed only when the arrays are pre-sliced to a shard. */
int vq_index_add_token_to_anchor_score(vq_index_builder*, const char* path, uint32_t num_tokens,
                                       const uint64_t* offsets /* [num_tokens+1] */,
                                       const uint32_t* anchors, const uint32_t* scores,
                                       const uint64_t* global_lens);

/* Any `IndexIdToParent<Output=u32>` store (src/persistence.rs:142-181):
 * ".tokens_to_text_id", ".text_id_to_anchor", ".parent_to_value_id",
 * ".anchor_to_text_id", ".value_id_to_anchor", ".value_id_to_parent",
 * ".text_id_to_token_ids".  Key k (k = key_base + row) maps to
 * values[offsets[row] .. offsets[row+1]]. */
int vq_index_add_key_value_store(vq_index_builder*, const char* path, uint32_t key_base,
                                 uint32_t num_keys, const uint64_t* offsets, const uint32_t* values);

/* "<field>.textindex.phrase_pair_to_anchor" — replaces
 * `PhrasePairToAnchor::get_values((u32,u32))` (src/persistence.rs:84-87,
 * src/indices/persistence_data_binary_search.rs:167-202).  Keys sorted by (t1,t2). */
int vq_index_add_phrase_pair_to_anchor(vq_index_builder*, const char* path, uint64_t num_pairs,
                                       const uint32_t* t1, const uint32_t* t2,
                                       const uint64_t* offsets /* [num_pairs+1] */,
                                       const uint32_t* anchors);

/* "<field>.boost_valid_to_value" — replaces `persistence.get_boost(path)`
 * (src/persistence.rs boost_valueid_to_value; read at src/search/boost.rs:490-494).
 * `present[row]` != 0 when key (key_base+row) has a value; value_bits = f32 bits. */
int vq_index_add_boost(vq_index_builder*, const char* path, uint32_t key_base, uint32_t num_keys,
                       const uint8_t* present /* may be NULL = all present */,
                       const uint32_t* value_bits);

/* Column metadata consulted on the query path: `is_anchor_identity_column`
 * (src/search/search_field.rs:474-480, src/search/boost.rs:60-66) and
 * `textindex_metadata.options.tokenize` (src/search/search_field.rs:650-658). */
int vq_index_set_column_meta(vq_index_builder*, const char* field, int is_anchor_identity_column, int tokenize);

/* Stage everything into HBM on `device` (padded segmented arrays) and return the
 * immutable index.  The builder stays valid and must still be freed. */
int vq_index_build(vq_index_builder*, int device, vq_index** out);
void vq_index_free(vq_index*);

/* Run this index's launches on an existing HIP stream (hipStream_t as void*);
 * NULL restores the index's own stream. */
int vq_index_set_stream(vq_index*, void* hip_stream);
/* Two caller streams: the scans of vq_search_batch_partial run on `scan_stream`, everything vq_merge_partials* launches (shard merge,
 * facet selection, result download) on `finish_stream`, which waits for the batch's scan through an event.  A caller that pipelines
 * batches puts its collective on `finish_stream` too (after making that stream wait for the scan, e.g. an event recorded on
 * `scan_stream` right after vq_search_batch_partial returned): the gather and merge of batch c then overlap the scan of batch c+1. */
int vq_index_set_streams(vq_index*, void* scan_stream, void* finish_stream);
/* Sharded deployments: a few requests need numbers that are sums over all shards before they can be compiled (result sizes of AND
 * operands, lengths of merged leaf lists — set_op.rs:388-393 orders an AND's score sum by them).  `fn(ctx, values, n)` must replace
 * values[0..n) by their sums over all ranks (an all-reduce; every rank calls it with the same n, in the same order) and return 0.
 * Without it a shard declines such requests with VQ_ERR_UNSUPPORTED.  Called from inside vq_search_batch_partial. */
typedef int (*vq_allreduce_u64_fn)(void* ctx, uint64_t* values, size_t n);
int vq_index_set_allreduce(vq_index*, vq_allreduce_u64_fn fn, void* ctx);
/* Bytes of HBM held by the staged image. */
uint64_t vq_index_device_bytes(const vq_index*);

/* --------------------------------------------------------------- requests
 *
 * `vq_request` == `search::Request` (src/search/request/mod.rs:15-87), parsed
 * from its serde-JSON rendering.  Unknown keys are ignored (as serde does). */
typedef struct vq_request vq_request;

int vq_request_parse(const char* json, size_t len, vq_request** out);
void vq_request_free(vq_request*);
/* What the parser understood: every field of search::Request (src/search/request/mod.rs:15-87, search_request.rs:6-179, boost_request.rs:4-33,
 * facet_request.rs:2-11) in declaration order, absent Options as null, `select` / `snippet_info` as "present" booleans, f32 values as their bit
 * patterns (u32).  Pinned by tests/golden/request_parse.json against an independent restatement of serde's rules.  Thread-local string. */
const char* vq_request_to_json(const vq_request*);
/* 1 when the request asks for facets (Request::facets, src/search/request/mod.rs:39, is a non-empty list), else 0: what decides whether a sharded
 * step exchanges histograms — taken from the PARSED request, the same on every rank. */
int vq_request_has_facets(const vq_request*);
/* str::to_lowercase as the dictionary side applies it (src/search/search_field.rs:284,312).  Returns the byte length written to `out`, or
 * (size_t)-1 when `cap` is too small.  Diagnostic: swept over every code point by tests/test_request_parse.py. */
size_t vq_debug_to_lowercase(const char* utf8, size_t len, char* out, size_t cap);
/* util::normalize_text (src/util.rs:11-29) as vq_highlight_json applies it to a part's terms; same conventions.  Diagnostic: compared with an
 * independent regex engine by tests/test_request_parse.py. */
size_t vq_debug_normalize_text(const char* utf8, size_t len, char* out, size_t cap);
/* The request compiler's id-list sort (value ids behind 1:n boost lists: ascending, duplicate-free, in place; returns the new length).  Diagnostic:
 * its three regimes are compared with an independent sort by tests/test_request_parse.py. */
size_t vq_debug_sort_unique_u32(uint32_t* ids, size_t n);
/* Compile a request against an index without launching anything (host-only): 0 = ready to scan, negative = a pre-pass would run first (-1 union /
 * locality jobs, -2 count pre-pass, -3 range jobs), otherwise the error code the search would return.  Diagnostic: the CPU sanitizer build (`make asan`) runs it over the
 * request fixtures; tools/compile_bench.py times it. */
int vq_debug_compile(const vq_index*, const vq_request*);

/* ---------------------------------------------------------------- results
 *
 * `vq_result` == `search::SearchResult` (src/search/result/search_result.rs:9-26). */
typedef struct vq_result vq_result;

uint64_t vq_result_num_hits(const vq_result*);
uint64_t vq_result_execution_time_ns(const vq_result*);
size_t vq_result_len(const vq_result*);            /* data.len() */
const uint32_t* vq_result_ids(const vq_result*);   /* data[i].id    */
const float* vq_result_scores(const vq_result*);   /* data[i].score */
size_t vq_result_num_facets(const vq_result*);
const char* vq_result_facet_field(const vq_result*, size_t facet);
size_t vq_result_facet_len(const vq_result*, size_t facet);
const char* vq_result_facet_value(const vq_result*, size_t facet, size_t i);
uint64_t vq_result_facet_count(const vq_result*, size_t facet, size_t i);
/* serde_json rendering of the SearchResult (ids/scores/facets/why_found_terms/why_found_info), for diffing. */
const char* vq_result_to_json(const vq_result*);
/* `SearchResult.why_found_terms` (src/search/result/search_result.rs:21-25, filled at src/search.rs:186 when the request says
 * `why_found: true`): per text index path the matched dictionary terms, as JSON {"<path>": ["term", ...]} (thread-local string;
 * the reference's map and list orders are unspecified). */
const char* vq_result_why_found_terms_json(const vq_result*);
/* `SearchResult.why_found_info` (src/search.rs:220-224 = get_why_found, src/search/why_found.rs:11-50; what to_documents hands out per hit when
 * the request has a `select`, search.rs:83-88): for `why_found: true` together with `select`, per returned anchor and searched field the
 * anchor's texts of the field that hold a matched token, highlighted like vq_highlight_json's snippets with the default snippet options — as JSON
 * {"<anchor id>": {"<field>": ["text with <b>hits</b>", ...]}} (thread-local string); "null" when the request did not ask.  `select` itself is not
 * looked at, as in search::search: reading the selected fields is to_documents' work on the caller's document store.  Host work on the final
 * window of hits (needs the fields' `.parent_to_value_id` and `.text_id_to_token_ids` stores); vq_search / vq_search_json / vq_search_batch
 * only — the flat and the sharded entry points answer VQ_ERR_UNSUPPORTED for such a request. */
const char* vq_result_why_found_info_json(const vq_result*);
/* `explain: true` (src/search/request/mod.rs:83-86; or `options.explain` on the single leaf of a request): the `Explain` records
 * (src/search/result/explain.rs:2-21) of the returned hits, i.e. `SearchResult.explain.get(&hit.id)` of src/search.rs:86,96 — a JSON array
 * parallel to the hits, each element null or the hit's records in serde's form ({"TermToAnchor":{"term_score":..,"anchor_score":..,
 * "final_score":..,"term_id":..}}, {"LevenshteinScore":{"score":..,"text_or_token_id":"..","term_id":..}}, {"OrSumOverDistinctTerms":..},
 * {"Boost":..}; floats printed with %.9g).  "null" when the request did not ask.  The scores inside the records are recomputed on the
 * device for the returned hits only (k_explain).  Declined: explain together with phrase_boosts or a 1:n boost, explain on some leaves
 * only, and explain on the flat / sharded paths.  Thread-local string. */
const char* vq_result_explain_json(const vq_result*);
void vq_result_free(vq_result*);

/* ---------------------------------------------------------------- suggest
 *
 * == search_field::suggest_multi(persistence, request) (src/search/search_field.rs:194-219) when `json` is a Request with
 * "suggest": [RequestSearchPart, ...] (+ top / skip), == search_field::suggest(persistence, &part) (:221-231) when it is a bare
 * RequestSearchPart.  Dictionary side only: the matched terms of every part (fuzzy / prefix scans run on the device, regex leaves on
 * the host), lower-cased texts, equal texts merged keeping the best score, ranked by score.  SuggestFieldResult = Vec<(String, Score, TermId)>. */
typedef struct vq_suggest_result vq_suggest_result;
int vq_suggest_json(const vq_index*, const char* json, size_t len, vq_suggest_result** out);
size_t vq_suggest_len(const vq_suggest_result*);
const char* vq_suggest_text(const vq_suggest_result*, size_t i);
float vq_suggest_score(const vq_suggest_result*, size_t i);
uint32_t vq_suggest_term_id(const vq_suggest_result*, size_t i);
void vq_suggest_free(vq_suggest_result*);

/* ---------------------------------------------------------------- highlight
 * == search_field::highlight(persistence, &mut part) (src/search/search_field.rs:233-245): `json` is a bare RequestSearchPart with
 * "snippet": true (+ optional "snippet_info", src/search/request/snippet_info.rs).  The terms are normalised (util::normalize_text), matched
 * against the field's dictionary (prefix / fuzzy scans run on the device), resolved to the texts that contain a matched token
 * (resolve_token_hits_to_text_id, :550-639) and every such text is returned as a snippet (highlight_document, src/highlight_field.rs:187-272)
 * with the best score among its matched tokens, ranked by score, cut by the part's top / skip.  Entry i: vq_suggest_text = the snippet,
 * vq_suggest_score, vq_suggest_term_id = the text id.  Where the reference panics (a hit without a snippet: an untokenised field, "snippet" not
 * set) the call returns VQ_ERR_INVALID_REQUEST.  Needs the field's tokens_to_text_id and text_id_to_token_ids stores. */
int vq_highlight_json(const vq_index*, const char* json, size_t len, vq_suggest_result** out);
/* == highlight_field::highlight_text(text, set, opt, tokenizer) (src/highlight_field.rs:92-146): `text` with the tokens that are in `terms` wrapped in
 * the snippet tags, cut into windows like vq_highlight_json's snippets — what why_found highlighting (highlight_on_original_document, :148-185) applies to
 * every text of a returned document with the field's terms from vq_result_why_found_terms_json.  `tokenized` = the field has a tokenizer (the default
 * separators of src/tokenizer/mod.rs:21-23); `snippet_info_json` may be NULL (DEFAULT_SNIPPETINFO); term_lens may be NULL (C strings).  Returns the
 * snippet's byte length — the bytes are in `out` when the length fits `cap`, otherwise call again with a larger buffer —, VQ_HIGHLIGHT_NONE when
 * nothing is highlighted (the reference's None), VQ_HIGHLIGHT_ERROR on an error (vq_last_error).  Host work; needs no index. */
#define VQ_HIGHLIGHT_NONE ((size_t)-1)
#define VQ_HIGHLIGHT_ERROR ((size_t)-2)
size_t vq_highlight_text(const char* text, size_t len, const char* const* terms, const size_t* term_lens, size_t n_terms, const char* snippet_info_json,
                         size_t json_len, int tokenized, char* out, size_t cap);

/* ----------------------------------------------------------------- search */

/* == search::search(request, &persistence) (src/search.rs:143-228). */
int vq_search(const vq_index*, const vq_request*, vq_result** out);
/* Same, from the request's JSON text. */
int vq_search_json(const vq_index*, const char* json, size_t len, vq_result** out);
/* Throughput path: n independent searches executed as one device batch
 * (the reference's equivalent is n concurrent `search()` calls from server
 * threads, server/rocket_server.rs:139-145).  out[i] receives request i's
 * result, or NULL with status[i] != VQ_OK.  Returns VQ_OK when the batch ran. */
int vq_search_batch(const vq_index*, const vq_request* const* requests, size_t n,
                    vq_result** out, int* status);

/* Same batch, flat output without per-result objects (facets are not reported through this entry):
 * request i's hits go to ids/scores[i * stride .. i * stride + counts[i]).  `stride` must be >= the
 * largest `top` of the batch. */
int vq_search_batch_flat(const vq_index*, const vq_request* const* requests, size_t n, size_t stride,
                         uint64_t* num_hits /* [n] */, uint32_t* counts /* [n] */,
                         uint32_t* ids /* [n*stride] */, float* scores /* [n*stride] */, int* status /* [n] */);

/* ------------------------------------------------- shard-partial interface
 *
 * New surface (the reference has no sharding): a shard returns
 * its exact local top-(top+skip) with global doc ids, its local hit count and
 * its facet histograms over value ids; `vq_merge_partials` turns the gathered
 * partials of all shards into the final results.  The gather itself is done by
 * the caller (RCCL all-gather of the packed buffers). */
typedef struct vq_partial_batch vq_partial_batch;

int vq_search_batch_partial(const vq_index*, const vq_request* const* requests, size_t n,
                            vq_partial_batch** out);
/* Packed, fixed-size representation of the partials (same size on every shard
 * for the same requests).  Resident in HBM: `device_ptr` can be handed to RCCL. */
size_t vq_partial_bytes(const vq_partial_batch*);
void* vq_partial_device_ptr(vq_partial_batch*);
/* The batch's facet histograms (u32 counts over value ids; 0 bytes without facet requests) live behind the all-gathered part
 * and are NOT part of it: the caller sums them over the shards in place with an all-reduce (ncclSum on `vq_partial_hist_bytes / 4`
 * u32 — SURVEY.md 8e: facet counts are additive) before `vq_merge_partials*`.  With one shard nothing needs to be done. */
size_t vq_partial_hist_bytes(const vq_partial_batch*);
void* vq_partial_hist_device_ptr(vq_partial_batch*);
/* Merge `num_shards` gathered packed buffers (device memory, shard-major,
 * each `vq_partial_bytes` long) into final results; facet counts are read from `local`'s (already summed) histograms. */
int vq_merge_partials(const vq_index*, vq_partial_batch* local, const void* gathered_device,
                      uint32_t num_shards, vq_result** out, int* status);
/* One scan ranks the best 1024 hits of a request.  A request whose top + skip reaches further (the reference has no limit: src/search/sort.rs:5-22,
 * src/search.rs:230-239) comes back from vq_merge_partials as its PAGE 0 — all 1024 ranked hits, top / skip not applied yet — with
 * vq_result_is_page() == 1.  The caller (every rank alike: the merged page is the same on all of them) sends vq_request_page_after(request, score and
 * id of the page's last hit) through partial -> all-gather -> merge again, appends the hits, and repeats while a page comes back full; then it
 * applies skip / top (apply_top_skip).  vq_search / vq_search_batch do this themselves on an unsharded index; the flat merges decline such requests. */
int vq_result_is_page(const vq_result*);
int vq_request_page_after(const vq_request*, float score, uint32_t id, vq_request** out);
/* Same merge with flat output (see vq_search_batch_flat). */
int vq_merge_partials_flat(const vq_index*, vq_partial_batch* local, const void* gathered_device, uint32_t num_shards, size_t stride,
                           uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status);
/* One collective per step while the step still runs as a pipeline of chunks (the host compiles chunk c+1 while the GPU scans chunk c):
 * every chunk is searched with `vq_search_batch_partial_at` into its own workspace `slot` (0 .. vq_partial_slots()-1) with its partial placed at
 * `arena_offset` of the index's partial arena (`vq_index_partial_arena_ptr`; offsets are multiples of 256, chunk c+1 starts at
 * arena_offset_c + round_up(vq_partial_total_bytes(chunk c), 256); VQ_ERR_UNSUPPORTED when the arena — 64 MiB — is too small).  The caller
 * all-gathers the arena's used prefix [0, S) ONCE, then merges every chunk with `vq_merge_partials_flat_strided(gathered + arena_offset_c,
 * num_shards, shard_stride = S)`.  Chunks with facet histograms still need their all-reduce (vq_partial_hist_*), so a caller would keep such
 * steps on the per-chunk path. */
int vq_search_batch_partial_at(const vq_index*, const vq_request* const* requests, size_t n, int slot, size_t arena_offset, vq_partial_batch** out);
int vq_partial_slots(void);
void* vq_index_partial_arena_ptr(const vq_index*);
size_t vq_partial_total_bytes(const vq_partial_batch*);
int vq_merge_partials_flat_strided(const vq_index*, vq_partial_batch* local, const void* gathered_device, uint32_t num_shards, size_t shard_stride,
                                   size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status);
void vq_partial_free(vq_partial_batch*);

/* ---- the sharded step inside the library (no counterpart in the reference, which does not shard).
 * One communicator per index / rank: rank 0 takes a unique id (`vq_comm_unique_id`, VQ_COMM_ID_BYTES bytes), hands it to the other ranks
 * by whatever means the host program has, every rank calls `vq_comm_init` (ncclCommInitRank on the index's device; RCCL is loaded at run
 * time).  From then on `vq_shard_step_flat` is one whole step on this rank: compile -> scans -> RCCL all-gather of the packed partials
 * (hit counts + top-k keys of every query; every rank must pass the SAME requests) and all-reduce of the facet histograms -> merge ->
 * one download; the communicator also serves the sums over the shards that some requests need before compilation (what
 * `vq_index_set_allreduce` asks a caller for).  `vq_shard_step_begin` returns as soon as the step's work is queued, `vq_shard_step_end`
 * delivers it: a caller that begins step i+1 before it ends step i keeps two steps in flight (the scans of one overlap the exchange and
 * merge of the other; at most two).  Results are those of `vq_search_batch_flat` on the unsharded index, bit for bit; requests whose
 * top + skip exceeds 1024 are declined here when the step has an exchange (page them through vq_merge_partials); an index that answers alone pages them itself.  On an index without a communicator the step is the same
 * pipeline without an exchange: begin / end then keep two batches in flight on an unsharded index.
 * `vq_comm_init_custom` takes the exchange from the caller instead (all-gather of `bytes_per_rank` bytes per rank into a rank-major
 * buffer; in-place sum of u32 counters — both on device memory, to be ordered on `hip_stream` or finished before they return): several
 * shards inside one process, tests. */
#define VQ_COMM_ID_BYTES 128
typedef struct vq_shard_step vq_shard_step;
typedef int (*vq_allgather_fn)(void* ctx, const void* local_device, void* gathered_device, size_t bytes_per_rank, void* hip_stream);
typedef int (*vq_allreduce_u32_fn)(void* ctx, void* inout_device, size_t count, void* hip_stream);
int vq_comm_unique_id(void* id_out);
int vq_comm_init(vq_index*, int nranks, int rank, const void* unique_id);
int vq_comm_init_custom(vq_index*, int nranks, int rank, vq_allgather_fn allgather, vq_allreduce_u32_fn allreduce_u32, void* ctx);
int vq_comm_destroy(vq_index*);
int vq_shard_step_begin(const vq_index*, const vq_request* const* requests, size_t n, vq_shard_step** out);
int vq_shard_step_end(vq_shard_step*, size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status); /* frees the step */
void vq_shard_step_free(vq_shard_step*); /* a step that is given up instead of ended */
int vq_shard_step_flat(const vq_index*, const vq_request* const* requests, size_t n, size_t stride, uint64_t* num_hits, uint32_t* counts,
                       uint32_t* ids, float* scores, int* status);

/* ------------------------------------------------------------ measurement */

/* Device time (ms, HIP events on the index's stream) and launch count of the
 * dominant scan kernel accumulated since the last call with reset != 0. */
int vq_profile_read(const vq_index*, int reset, double* scan_kernel_ms, uint64_t* scan_launches,
                    uint64_t* algorithmic_bytes);
/* Enable (1) / disable (0) the HIP-event bracketing used by vq_profile_read / vq_profile_json. */
int vq_profile_enable(vq_index*, int on);
/* Per-kernel accounting since the last reset, as JSON (thread-local string, valid until the next call on this thread):
 * {"batches": n, "kernels": {"<kernel>": {"ms", "launches", "layout_bytes", "algorithmic_bytes", "queries", "scan"}}}.
 * ms = device time between HIP events recorded around the launch on its stream; layout_bytes = the bytes THIS data layout has to
 * move for those launches (bitmap words of dense lists, 4 B per id of scattered lists, 6 B per streamed posting, the bytes of
 * per-hit gathers as counted by the kernels, 8 B per key written) — the roofline numerator; algorithmic_bytes = SURVEY.md 8(d)'s
 * posting-streaming accounting, kept for comparison. */
const char* vq_profile_json(const vq_index*, int reset);

/* ------------------------------------------------------------- self-checks (tests) */

/* Number of f16 inputs for which the kernels' fast `a / 100.0f` differs from the correctly rounded division (must be 0);
 * 0xFFFFFFFF when no device is available. */
uint32_t vq_debug_div100_mismatches(void);

/* Requests of this index that were run a second time on the exact kernels because the short cut of a speculative one could not be confirmed:
 * an OR of the shape k_scan_probe_or takes ranks only the docs that hold its rarest term and is confirmed by the k-th key it returns. */
uint64_t vq_index_speculative_reruns(const vq_index*);

/* The facet selection kernels (k_facet_select / k_facet_select_wide: facet.rs:19-23, count descending; ties by value id ascending) on a
 * caller's histogram of `num_values` counts, placed `misalign` (0-3) counters behind a 16-byte boundary as inside a batch's histogram area:
 * writes the best min(top, non-zero counts) entries, returns their number, -1 on failure. */
int vq_debug_facet_select(const uint32_t* hist, uint32_t num_values, uint32_t top, uint32_t misalign, uint32_t* out_values, uint32_t* out_counts);

const char* vq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VELOCI_AMD_H */
