// extern "C" boundary (include/veloci_amd.h).  No torch types, no exceptions across the ABI.
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/veloci_amd.h"
#include <functional>
#include <future>

#include <dlfcn.h>
#include <unistd.h>

#include "engine.hpp"
#include "text.hpp"

using namespace vq;
using vqreq::Request;
using vqreq::VelociError;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
template <class F>
int guard(F f) {
    try {
        f();
        g_err.clear();
        return VQ_OK;
    } catch (const VelociError& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(VQ_ERR_DEVICE, "out of host memory");
    } catch (const std::exception& e) {
        return fail(VQ_ERR_INVALID_ARGUMENT, e.what());
    }
}

void json_f32(std::string& out, float f) {
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.9g", double(f));
    out += buf;
    if (!std::strpbrk(buf, ".eEni")) out += ".0";
}
}  // namespace

struct vq_index_builder {
    IndexBuilder b;
};
struct vq_index {
    std::unique_ptr<Index> idx;
};
struct vq_request {
    Request req;
};
struct vq_result {
    Result r;
};
struct vq_partial_batch {
    std::unique_ptr<PartialBatch> pb;
};
struct vq_suggest_result {
    std::vector<SuggestEntry> e;
};

static void explain_json(std::string& s, const Result& R) {
    s += '[';
    for (size_t i = 0; i < R.explain.size(); ++i) {
        if (i) s += ',';
        s += R.explain[i].first ? explain_records_json(R.explain[i].second) : std::string("null");
    }
    s += ']';
}
static void why_found_terms_json(std::string& s, const std::map<std::string, std::vector<std::string>>& m) {
    s += '{';
    bool first = true;
    for (auto& kv : m) {
        if (!first) s += ',';
        first = false;
        vqjson::escape_to(s, kv.first);
        s += ":[";
        for (size_t i = 0; i < kv.second.size(); ++i) {
            if (i) s += ',';
            vqjson::escape_to(s, kv.second[i]);
        }
        s += ']';
    }
    s += '}';
}

static void why_found_info_json(std::string& s, const Result& R) {  // {"<anchor id>": {"<field>": ["highlighted text", ...]}}
    s += '{';
    bool first = true;
    for (auto& [anchor, fields] : R.why_found_info) {
        if (!first) s += ',';
        first = false;
        s += '"' + std::to_string(anchor) + "\":";
        why_found_terms_json(s, fields);
    }
    s += '}';
}

// ---- canonical dump of a parsed request: every field of search::Request in declaration order, absent options as null, f32 values as their bit
// patterns — what the parser understood, for the golden request-parse fixtures (tests/golden/request_parse.json)
namespace {
void dump_f32(std::string& s, float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    s += std::to_string(b);
}
template <class T, class F>
void dump_opt(std::string& s, const std::optional<T>& v, F f) {
    if (v) f(s, *v);
    else s += "null";
}
void dump_usize(std::string& s, size_t v) { s += std::to_string(v); }
void dump_bool(std::string& s, bool v) { s += v ? "true" : "false"; }
void dump_str(std::string& s, const std::string& v) { vqjson::escape_to(s, v); }
void dump_boost_part(std::string& s, const vqreq::RequestBoostPart& b) {
    static const char* const names[] = {"Log2", "Log10", "Multiply", "Add", "Replace"};
    s += "{\"path\":";
    dump_str(s, b.path);
    s += ",\"boost_fun\":";
    if (b.boost_fun) s += std::string("\"") + names[int(*b.boost_fun)] + "\"";
    else s += "null";
    s += ",\"param\":";
    dump_opt(s, b.param, dump_f32);
    s += ",\"skip_when_score\":";
    if (b.skip_when_score) {
        s += '[';
        for (size_t i = 0; i < b.skip_when_score->size(); ++i) {
            if (i) s += ',';
            dump_f32(s, (*b.skip_when_score)[i]);
        }
        s += ']';
    } else s += "null";
    s += ",\"expression\":";
    dump_opt(s, b.expression, dump_str);
    s += '}';
}
void dump_boost_list(std::string& s, const std::optional<std::vector<vqreq::RequestBoostPart>>& v) {
    if (!v) {
        s += "null";
        return;
    }
    s += '[';
    for (size_t i = 0; i < v->size(); ++i) {
        if (i) s += ',';
        dump_boost_part(s, (*v)[i]);
    }
    s += ']';
}
void dump_options(std::string& s, const std::optional<vqreq::SearchRequestOptions>& o) {
    if (!o) {
        s += "null";
        return;
    }
    s += "{\"explain\":";
    dump_bool(s, o->explain);
    s += ",\"top\":";
    dump_opt(s, o->top, dump_usize);
    s += ",\"skip\":";
    dump_opt(s, o->skip, dump_usize);
    s += ",\"boost\":";
    dump_boost_list(s, o->boost);
    s += '}';
}
void dump_part(std::string& s, const vqreq::RequestSearchPart& p) {
    s += "{\"path\":";
    dump_str(s, p.path);
    s += ",\"terms\":[";
    for (size_t i = 0; i < p.terms.size(); ++i) {
        if (i) s += ',';
        dump_str(s, p.terms[i]);
    }
    s += "],\"levenshtein_distance\":";
    if (p.levenshtein_distance) s += std::to_string(*p.levenshtein_distance);
    else s += "null";
    s += ",\"starts_with\":";
    dump_bool(s, p.starts_with);
    s += ",\"is_regex\":";
    dump_bool(s, p.is_regex);
    s += ",\"token_value\":";
    if (p.token_value) dump_boost_part(s, *p.token_value);
    else s += "null";
    s += ",\"boost\":";
    dump_opt(s, p.boost, dump_f32);
    s += ",\"ignore_case\":";
    dump_opt(s, p.ignore_case, dump_bool);
    s += ",\"snippet\":";
    dump_opt(s, p.snippet, dump_bool);
    s += ",\"snippet_info\":";
    dump_bool(s, p.has_snippet_info);
    s += ",\"top\":";
    dump_opt(s, p.top, dump_usize);
    s += ",\"skip\":";
    dump_opt(s, p.skip, dump_usize);
    s += ",\"options\":";
    dump_options(s, p.options);
    s += '}';
}
void dump_search_request(std::string& s, const vqreq::SearchRequest& r) {
    if (r.kind == vqreq::SearchRequest::Search) {
        s += "{\"search\":";
        dump_part(s, r.part);
        s += '}';
        return;
    }
    s += r.kind == vqreq::SearchRequest::Or ? "{\"or\":{\"queries\":[" : "{\"and\":{\"queries\":[";
    for (size_t i = 0; i < r.tree.queries.size(); ++i) {
        if (i) s += ',';
        dump_search_request(s, r.tree.queries[i]);
    }
    s += "],\"options\":";
    dump_options(s, r.tree.options);
    s += "}}";
}
void dump_parts(std::string& s, const std::optional<std::vector<vqreq::RequestSearchPart>>& v) {
    if (!v) {
        s += "null";
        return;
    }
    s += '[';
    for (size_t i = 0; i < v->size(); ++i) {
        if (i) s += ',';
        dump_part(s, (*v)[i]);
    }
    s += ']';
}
}  // namespace

extern "C" {

const char* vq_last_error(void) { return g_err.c_str(); }
const char* vq_version(void) { return "veloci_amd 0.3 (gfx950)"; }
/* self-check (tests): inputs for which the kernels' fast a/100 differs from the correctly rounded division, over all f16 values */
uint32_t vq_debug_div100_mismatches(void) { return vq::debug_div100_mismatches(); }
/* tests, tools: requests that ran a second time because a speculative route's result could not be confirmed (k_scan_probe_or) */
uint64_t vq_index_speculative_reruns(const vq_index* index) { return index ? index->idx->or_reruns.load() : 0; }
/* self-check (tests): the facet top-`top` kernels on a caller's histogram */
int vq_debug_facet_select(const uint32_t* hist, uint32_t num_values, uint32_t top, uint32_t misalign, uint32_t* out_values, uint32_t* out_counts) {
    if (!hist || !out_values || !out_counts) return -1;
    return vq::debug_facet_select(hist, num_values, top, misalign, out_values, out_counts);
}

// ------------------------------------------------------------------ index
vq_index_builder* vq_index_builder_new(uint32_t num_anchors, uint32_t doc_lo, uint32_t doc_hi) {
    auto* b = new vq_index_builder();
    b->b.num_anchors = num_anchors;
    b->b.doc_lo = doc_lo;
    b->b.doc_hi = doc_hi;
    return b;
}
void vq_index_builder_free(vq_index_builder* b) { delete b; }

int vq_index_add_fst(vq_index_builder* b, const char* path, uint32_t num_terms, const uint8_t* term_bytes, const uint64_t* term_offsets) {
    return guard([&] {
        if (!b || !path || (num_terms && (!term_bytes || !term_offsets))) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_fst: null argument");
        HostFst f;
        f.terms.reserve(num_terms);
        for (uint32_t i = 0; i < num_terms; ++i) {
            f.terms.emplace_back(reinterpret_cast<const char*>(term_bytes) + term_offsets[i], size_t(term_offsets[i + 1] - term_offsets[i]));
            if (i && !(f.terms[i - 1] < f.terms[i])) throw VelociError(VQ_ERR_INVALID_ARGUMENT, std::string("terms must be bytewise sorted and unique: ") + path);
        }
        b->b.fst[path] = std::move(f);
    });
}

int vq_index_add_token_to_anchor_score(vq_index_builder* b, const char* path, uint32_t num_tokens, const uint64_t* offsets, const uint32_t* anchors,
                                       const uint32_t* scores, const uint64_t* global_lens) {
    return guard([&] {
        if (!b || !path || !offsets) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_token_to_anchor_score: null argument");
        HostPostings p;
        p.offsets.assign(offsets, offsets + num_tokens + 1);
        const uint64_t n = offsets[num_tokens];
        if (n && (!anchors || !scores)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_token_to_anchor_score: null arrays");
        p.anchors.assign(anchors, anchors + n);
        p.scores.assign(scores, scores + n);
        for (uint32_t t = 0; t < num_tokens; ++t)
            for (uint64_t i = offsets[t] + 1; i < offsets[t + 1]; ++i)
                if (!(anchors[i - 1] < anchors[i])) throw VelociError(VQ_ERR_INVALID_ARGUMENT, std::string("posting lists must be ascending and unique: ") + path);
        if (global_lens) p.global_lens.assign(global_lens, global_lens + num_tokens);
        b->b.postings[path] = std::move(p);
    });
}

int vq_index_add_key_value_store(vq_index_builder* b, const char* path, uint32_t key_base, uint32_t num_keys, const uint64_t* offsets,
                                 const uint32_t* values) {
    return guard([&] {
        if (!b || !path || !offsets) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_key_value_store: null argument");
        HostKV k;
        k.key_base = key_base;
        k.offsets.assign(offsets, offsets + num_keys + 1);
        const uint64_t n = offsets[num_keys];
        if (n && !values) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_key_value_store: null values");
        k.values.assign(values, values + n);
        b->b.kv[path] = std::move(k);
    });
}

int vq_index_add_phrase_pair_to_anchor(vq_index_builder* b, const char* path, uint64_t num_pairs, const uint32_t* t1, const uint32_t* t2,
                                       const uint64_t* offsets, const uint32_t* anchors) {
    return guard([&] {
        if (!b || !path || !offsets) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_phrase_pair_to_anchor: null argument");
        HostPhrase p;
        p.t1.assign(t1, t1 + num_pairs);
        p.t2.assign(t2, t2 + num_pairs);
        p.offsets.assign(offsets, offsets + num_pairs + 1);
        p.anchors.assign(anchors, anchors + offsets[num_pairs]);
        b->b.phrase[path] = std::move(p);
    });
}

int vq_index_add_boost(vq_index_builder* b, const char* path, uint32_t key_base, uint32_t num_keys, const uint8_t* present, const uint32_t* value_bits) {
    return guard([&] {
        if (!b || !path || (num_keys && !value_bits)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_add_boost: null argument");
        HostBoost h;
        h.key_base = key_base;
        if (present) h.present.assign(present, present + num_keys);
        h.bits.assign(value_bits, value_bits + num_keys);
        b->b.boost[path] = std::move(h);
    });
}

int vq_index_set_column_meta(vq_index_builder* b, const char* field, int is_anchor_identity_column, int tokenize) {
    return guard([&] {
        if (!b || !field) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_set_column_meta: null argument");
        ColumnMeta m;
        m.is_anchor_identity_column = is_anchor_identity_column != 0;
        m.tokenize = tokenize != 0;
        b->b.columns[field] = m;
    });
}

int vq_index_build(vq_index_builder* b, int device, vq_index** out) {
    return guard([&] {
        if (!b || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_build: null argument");
        *out = nullptr;
        auto idx = build_index(b->b, device);
        auto* h = new vq_index();
        h->idx = std::move(idx);
        *out = h;
    });
}
void vq_index_free(vq_index* i) { delete i; }

int vq_index_set_stream(vq_index* i, void* hip_stream) {
    return guard([&] {
        if (!i) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_set_stream: null index");
        std::vector<std::unique_lock<std::mutex>> quiet;  // no batch in flight
        for (auto& w : i->idx->ws) quiet.emplace_back(w.mu);
        i->idx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : i->idx->own_stream;
        i->idx->fin_stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : i->idx->own_fin_stream;
    });
}
int vq_index_set_streams(vq_index* i, void* scan_stream, void* finish_stream) {
    return guard([&] {
        if (!i) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_set_streams: null index");
        if (!scan_stream || !finish_stream) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_set_streams: null stream (vq_index_set_stream(index, NULL) restores the index's own)");
        std::vector<std::unique_lock<std::mutex>> quiet;  // no batch in flight
        for (auto& w : i->idx->ws) quiet.emplace_back(w.mu);
        i->idx->stream = static_cast<hipStream_t>(scan_stream);
        i->idx->fin_stream = static_cast<hipStream_t>(finish_stream);
    });
}
int vq_index_set_allreduce(vq_index* i, vq_allreduce_u64_fn fn, void* ctx) {
    return guard([&] {
        if (!i) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_index_set_allreduce: null index");
        std::vector<std::unique_lock<std::mutex>> quiet;  // no batch in flight
        for (auto& w : i->idx->ws) quiet.emplace_back(w.mu);
        i->idx->allreduce_fn = fn;
        i->idx->allreduce_ctx = ctx;
    });
}
uint64_t vq_index_device_bytes(const vq_index* i) { return i ? i->idx->device_bytes : 0; }

// ------------------------------------------------------------------ requests
int vq_request_parse(const char* json, size_t len, vq_request** out) {
    return guard([&] {
        if (!json || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_request_parse: null argument");
        *out = nullptr;
        auto* r = new vq_request();
        try {
            r->req = vqreq::request_from_json_text(json, len);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
    });
}
int vq_request_has_facets(const vq_request* r) { return r && r->req.facets && !r->req.facets->empty() ? 1 : 0; }
const char* vq_request_to_json(const vq_request* r) {
    thread_local std::string s;
    s.clear();
    if (!r) return "null";
    const Request& q = r->req;
    s += "{\"search_req\":";
    if (q.search_req) dump_search_request(s, *q.search_req);
    else s += "null";
    s += ",\"suggest\":";
    dump_parts(s, q.suggest);
    s += ",\"boost\":";
    dump_boost_list(s, q.boost);
    s += ",\"boost_term\":";
    dump_parts(s, q.boost_term);
    s += ",\"facets\":";
    if (q.facets) {
        s += '[';
        for (size_t i = 0; i < q.facets->size(); ++i) {
            if (i) s += ',';
            s += "{\"field\":";
            dump_str(s, (*q.facets)[i].field);
            s += ",\"top\":";
            dump_opt(s, (*q.facets)[i].top, dump_usize);
            s += '}';
        }
        s += ']';
    } else s += "null";
    s += ",\"phrase_boosts\":";
    if (q.phrase_boosts) {
        s += '[';
        for (size_t i = 0; i < q.phrase_boosts->size(); ++i) {
            if (i) s += ',';
            s += "{\"search1\":";
            dump_part(s, (*q.phrase_boosts)[i].search1);
            s += ",\"search2\":";
            dump_part(s, (*q.phrase_boosts)[i].search2);
            s += '}';
        }
        s += ']';
    } else s += "null";
    s += ",\"select\":";
    dump_bool(s, q.has_select);
    s += ",\"filter\":";
    if (q.filter) dump_search_request(s, *q.filter);
    else s += "null";
    s += ",\"top\":";
    dump_opt(s, q.top, dump_usize);
    s += ",\"skip\":";
    dump_opt(s, q.skip, dump_usize);
    s += ",\"why_found\":";
    dump_bool(s, q.why_found);
    s += ",\"text_locality\":";
    dump_bool(s, q.text_locality);
    s += ",\"explain\":";
    dump_bool(s, q.explain);
    s += '}';
    return s.c_str();
}
// Unicode lowercasing as the dictionary side applies it (str::to_lowercase, search_field.rs:284,312): for the code-point sweep against an
// independent implementation (tests/test_request_parse.py).  Returns the length, or (size_t)-1 when `cap` is too small.
size_t vq_debug_to_lowercase(const char* utf8, size_t len, char* out, size_t cap) {
    const std::string low = vqtext::to_lower_utf8(std::string(utf8 ? utf8 : "", utf8 ? len : 0));
    if (low.size() > cap) return size_t(-1);
    std::memcpy(out, low.data(), low.size());
    return low.size();
}
// util::normalize_text (src/util.rs:11-29) as vq_highlight_json applies it to the terms: checked against an independent regex engine by
// tests/test_request_parse.py.  Returns the length, or (size_t)-1 when `cap` is too small.
size_t vq_debug_normalize_text(const char* utf8, size_t len, char* out, size_t cap) {
    const std::string norm = vqtext::normalize_text(std::string(utf8 ? utf8 : "", utf8 ? len : 0));
    if (norm.size() > cap) return size_t(-1);
    std::memcpy(out, norm.data(), norm.size());
    return norm.size();
}
// The request compiler's id-list sort (ascending, duplicate-free, in place; returns the new length): a bitmap over dense id spans, radix passes over
// sparse ones, std::sort for short lists — swept against numpy by tests/test_request_parse.py.
size_t vq_debug_sort_unique_u32(uint32_t* ids, size_t n) { return ids || !n ? vq::debug_sort_unique(ids, n) : 0; }
// Compile `request` against `index` without launching anything: 0 when the query is ready to scan, negative when a pre-pass would run first
// (-1 union / locality jobs, -2 count pre-pass, -3 range jobs), or the error code the search would return (message in vq_last_error).  Host-only work —
// what the CPU sanitizer build exercises, and what tools/compile_bench.py times.
int vq_debug_compile(const vq_index* index, const vq_request* request) {
    int status = 0;
    const int rc = guard([&] {
        if (!index || !request) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_debug_compile: null argument");
        vq::CompiledQuery cq = vq::compile_query(*index->idx, request->req, nullptr, nullptr, nullptr, nullptr, nullptr);
        status = cq.status;
        if (cq.status != 0) g_err = cq.error;
    });
    return rc != 0 ? rc : status;
}
void vq_request_free(vq_request* r) { delete r; }
// The continuation of `request` behind the ranked hit (score, id): what a caller of the sharded partial / merge path sends for the next page of a
// request whose top + skip reaches beyond one scan's ranking (vq_result_is_page).
int vq_request_page_after(const vq_request* request, float score, uint32_t id, vq_request** out) {
    return guard([&] {
        if (!request || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_request_page_after: null argument");
        auto* r = new vq_request();
        r->req = vq::page_request_after(request->req, score, id);
        *out = r;
    });
}

// ------------------------------------------------------------------ results
uint64_t vq_result_num_hits(const vq_result* r) { return r->r.num_hits; }
int vq_result_is_page(const vq_result* r) { return r->r.deep ? 1 : 0; }
uint64_t vq_result_execution_time_ns(const vq_result* r) { return r->r.execution_time_ns; }
size_t vq_result_len(const vq_result* r) { return r->r.ids.size(); }
const uint32_t* vq_result_ids(const vq_result* r) { return r->r.ids.data(); }
const float* vq_result_scores(const vq_result* r) { return r->r.scores.data(); }
size_t vq_result_num_facets(const vq_result* r) { return r->r.facets.size(); }
const char* vq_result_facet_field(const vq_result* r, size_t f) { return r->r.facets[f].field.c_str(); }
size_t vq_result_facet_len(const vq_result* r, size_t f) { return r->r.facets[f].entries.size(); }
const char* vq_result_facet_value(const vq_result* r, size_t f, size_t i) { return r->r.facets[f].entries[i].first.c_str(); }
uint64_t vq_result_facet_count(const vq_result* r, size_t f, size_t i) { return r->r.facets[f].entries[i].second; }
const char* vq_result_to_json(const vq_result* r) {
    std::string& s = r->r.json;
    s.clear();
    s += "{\"execution_time_ns\":" + std::to_string(r->r.execution_time_ns) + ",\"num_hits\":" + std::to_string(r->r.num_hits) + ",\"data\":[";
    for (size_t i = 0; i < r->r.ids.size(); ++i) {
        if (i) s += ',';
        s += "{\"id\":" + std::to_string(r->r.ids[i]) + ",\"score\":";
        json_f32(s, r->r.scores[i]);
        s += '}';
    }
    s += "],\"ids\":[]";
    if (r->r.has_facets) {
        s += ",\"facets\":{";
        for (size_t f = 0; f < r->r.facets.size(); ++f) {
            if (f) s += ',';
            vqjson::escape_to(s, r->r.facets[f].field);
            s += ":[";
            for (size_t i = 0; i < r->r.facets[f].entries.size(); ++i) {
                if (i) s += ',';
                s += '[';
                vqjson::escape_to(s, r->r.facets[f].entries[i].first);
                s += ',' + std::to_string(r->r.facets[f].entries[i].second) + ']';
            }
            s += ']';
        }
        s += '}';
    }
    if (r->r.has_explain) {
        s += ",\"explain\":";
        explain_json(s, r->r);
    }
    if (!r->r.why_found_terms.empty()) {
        s += ",\"why_found_terms\":";
        why_found_terms_json(s, r->r.why_found_terms);
    }
    if (r->r.why_found_plan) {
        s += ",\"why_found_info\":";
        why_found_info_json(s, r->r);
    }
    s += '}';
    return s.c_str();
}
const char* vq_result_explain_json(const vq_result* r) {
    thread_local std::string s;
    s.clear();
    if (!r->r.has_explain) return "null";
    explain_json(s, r->r);
    return s.c_str();
}
const char* vq_result_why_found_terms_json(const vq_result* r) {
    thread_local std::string s;
    s.clear();
    why_found_terms_json(s, r->r.why_found_terms);
    return s.c_str();
}
const char* vq_result_why_found_info_json(const vq_result* r) {
    thread_local std::string s;
    s.clear();
    if (!r->r.why_found_plan) return "null";
    why_found_info_json(s, r->r);
    return s.c_str();
}
void vq_result_free(vq_result* r) { delete r; }

// ------------------------------------------------------------------ suggest
int vq_suggest_json(const vq_index* index, const char* json, size_t len, vq_suggest_result** out) {
    return guard([&] {
        if (!index || !json || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_suggest_json: null argument");
        *out = nullptr;
        Request req;
        try {
            vqjson::Value v = vqjson::parse(json, len);
            if (v.is_object() && v.get("suggest")) req = vqreq::request_from_json(v);
            else {  // a bare RequestSearchPart: search_field::suggest (:221-231), top / skip are the part's
                vqreq::RequestSearchPart part = vqreq::search_part_from_json(v);
                req.suggest = std::vector<vqreq::RequestSearchPart>{part};
                req.top = part.top;
                req.skip = part.skip;
            }
        } catch (const vqjson::ParseError& e) {
            throw VelociError(VQ_ERR_JSON, std::string("JsonError: ") + e.what());
        }
        auto* r = new vq_suggest_result();
        try {
            r->e = run_suggest(*index->idx, req);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
    });
}
int vq_highlight_json(const vq_index* index, const char* json, size_t len, vq_suggest_result** out) {
    return guard([&]() {
        if (!index || !json || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_highlight_json: null argument");
        *out = nullptr;
        vqreq::RequestSearchPart part;
        try {
            part = vqreq::search_part_from_json(vqjson::parse(json, len));
        } catch (const vqjson::ParseError& e) {
            throw VelociError(VQ_ERR_JSON, std::string("JsonError: ") + e.what());
        }
        auto* r = new vq_suggest_result();
        try {
            r->e = run_highlight(*index->idx, std::move(part));
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
    });
}
// highlight_text (highlight_field.rs:92-146).  Returns the snippet's byte length (written to `out` when it fits `cap`), VQ_HIGHLIGHT_NONE when there is
// nothing to highlight, VQ_HIGHLIGHT_ERROR on an error (vq_last_error); a length above `cap` means: call again with a larger buffer.
size_t vq_highlight_text(const char* text, size_t len, const char* const* terms, const size_t* term_lens, size_t n_terms, const char* snippet_info_json, size_t json_len,
                         int tokenized, char* out, size_t cap) {
    size_t result = size_t(-2);
    guard([&] {
        if ((!text && len) || (!terms && n_terms) || (!out && cap)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_highlight_text: null argument");
        vqreq::SnippetInfo opt;
        if (snippet_info_json && json_len) {
            try {
                opt = vqreq::snippet_info_from_json(vqjson::parse(snippet_info_json, json_len));
            } catch (const vqjson::ParseError& e) {
                throw VelociError(VQ_ERR_JSON, std::string("JsonError: ") + e.what());
            }
        }
        std::vector<std::string> set;
        for (size_t i = 0; i < n_terms; ++i) set.emplace_back(terms[i] ? terms[i] : "", terms[i] ? (term_lens ? term_lens[i] : std::strlen(terms[i])) : 0);
        const std::optional<std::string> sn = vq::highlight_text(std::string(text ? text : "", len), set, opt, tokenized != 0);
        if (!sn) {
            result = size_t(-1);
            return;
        }
        if (sn->size() <= cap) std::memcpy(out, sn->data(), sn->size());
        result = sn->size();
    });
    return result;
}
size_t vq_suggest_len(const vq_suggest_result* r) { return r->e.size(); }
const char* vq_suggest_text(const vq_suggest_result* r, size_t i) { return r->e[i].text.c_str(); }
float vq_suggest_score(const vq_suggest_result* r, size_t i) { return r->e[i].score; }
uint32_t vq_suggest_term_id(const vq_suggest_result* r, size_t i) { return r->e[i].term_id; }
void vq_suggest_free(vq_suggest_result* r) { delete r; }

// ------------------------------------------------------------------ search
// does the batch have pre-passes (dictionary scans of fuzzy / prefix leaves, then unions, range jobs ...)?  Their syncs make the host side of a
// batch long; such batches are cut in two and run on two host threads (a sample decides: it is about the batch's character)
static bool batch_has_prepasses(const std::vector<const Request*>& reqs) {
    std::function<bool(const vqreq::SearchRequest&)> scans = [&](const vqreq::SearchRequest& r) {
        if (r.kind == vqreq::SearchRequest::Search) return vq::needs_dictionary_scan(r.part);
        for (auto& q : r.tree.queries)
            if (scans(q)) return true;
        return false;
    };
    for (size_t i = 0; i < reqs.size(); i += 16)
        if (reqs[i] && reqs[i]->search_req && scans(*reqs[i]->search_req)) return true;
    return false;
}
static int run_batch(const vq_index* index, const vq_request* const* requests, size_t n, vq_result** out, int* status, std::string* first_error) {
    std::vector<const Request*> reqs(n);
    for (size_t i = 0; i < n; ++i) reqs[i] = requests[i] ? &requests[i]->req : nullptr;
    std::vector<std::unique_ptr<Result>> results(n);
    std::vector<int> st(n, 0);
    std::vector<std::string> errs(n);
    auto run_range = [&](size_t b, size_t e) {
        std::vector<std::unique_ptr<Result>> r;
        std::vector<int> s2;
        std::vector<std::string> e2;
        {
            auto pb = run_partial(*index->idx, reqs.data() + b, e - b);
            finish_batch(*index->idx, *pb, nullptr, 1, r, s2, e2);
        }  // (the batch's workspace is free again: deep requests scan on)
        complete_deep_requests(*index->idx, reqs.data() + b, e - b, r, s2, e2);
        complete_explain_requests(*index->idx, r, s2, e2);
        complete_why_found_requests(*index->idx, r, s2, e2);
        for (size_t i = b; i < e; ++i) {
            results[i] = std::move(r[i - b]);
            st[i] = s2[i - b];
            errs[i] = std::move(e2[i - b]);
        }
    };
    // A batch with pre-passes waits for the device several times before its scan starts (dictionary scans, unions, range counts): cut in
    // two and run on two host threads, one half computes while the other waits (VQ_BATCH_HALVES=0: one piece)
    static const bool halves = !(std::getenv("VQ_BATCH_HALVES") && std::atoi(std::getenv("VQ_BATCH_HALVES")) == 0);
    if (halves && n >= 128 && kWorkspaces >= 2 && batch_has_prepasses(reqs)) {
        auto other = std::async(std::launch::async, run_range, n / 2, n);
        try {
            run_range(0, n / 2);
        } catch (...) {
            other.wait();
            throw;
        }
        other.get();
    } else run_range(0, n);
    for (size_t i = 0; i < n; ++i) {
        out[i] = nullptr;
        if (status) status[i] = st[i];
        if (st[i] == 0) {
            auto* r = new vq_result();
            r->r = std::move(*results[i]);
            out[i] = r;
        } else if (first_error && first_error->empty()) *first_error = errs[i];
    }
    return 0;
}

int vq_search(const vq_index* index, const vq_request* request, vq_result** out) {
    int st = 0;
    std::string err;
    int rc = guard([&] {
        if (!index || !request || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search: null argument");
        *out = nullptr;
        const vq_request* arr[1] = {request};
        run_batch(index, arr, 1, out, &st, &err);
    });
    if (rc != VQ_OK) return rc;
    if (st != 0) return fail(st, err);
    return VQ_OK;
}

int vq_search_json(const vq_index* index, const char* json, size_t len, vq_result** out) {
    vq_request* req = nullptr;
    int rc = vq_request_parse(json, len, &req);
    if (rc != VQ_OK) return rc;
    rc = vq_search(index, req, out);
    vq_request_free(req);
    return rc;
}

int vq_search_batch(const vq_index* index, const vq_request* const* requests, size_t n, vq_result** out, int* status) {
    return guard([&] {
        if (!index || (n && (!requests || !out))) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search_batch: null argument");
        std::string err;
        run_batch(index, requests, n, out, status, &err);
        if (!err.empty()) g_err = err;
    });
}

// Explain records and why_found_info (why_found with select) are produced by vq_search / vq_search_json / vq_search_batch: the flat outputs have no
// place for them, and on the sharded path a rank only holds the postings and the texts of its own docs.
static void decline_explain(std::vector<std::unique_ptr<Result>>& results, std::vector<int>& st, std::vector<std::string>& errs, const char* where) {
    for (size_t i = 0; i < results.size(); ++i)
        if (st[i] == 0 && results[i] && (results[i]->explain_plan || results[i]->why_found_plan)) {
            st[i] = VQ_ERR_UNSUPPORTED;
            errs[i] = std::string("unsupported on the MI355X query path: ") + (results[i]->explain_plan ? "explain on " : "why_found with select on ") + where;
            results[i].reset();
        }
}
static void copy_flat(const std::vector<std::unique_ptr<Result>>& results, const std::vector<int>& st, const std::vector<std::string>& errs, size_t base,
                      size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status) {
    for (size_t k = 0; k < results.size(); ++k) {
        const size_t i = base + k;
        if (status) status[i] = st[k];
        num_hits[i] = 0;
        counts[i] = 0;
        if (st[k] != 0) {
            if (g_err.empty()) g_err = errs[k];
            continue;
        }
        const Result& r = *results[k];
        if (r.ids.size() > stride) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "flat output: stride smaller than a request's top");
        num_hits[i] = r.num_hits;
        counts[i] = uint32_t(r.ids.size());
        std::memcpy(ids + i * stride, r.ids.data(), r.ids.size() * 4);
        std::memcpy(scores + i * stride, r.scores.data(), r.scores.size() * 4);
    }
}

int vq_search_batch_flat(const vq_index* index, const vq_request* const* requests, size_t n, size_t stride, uint64_t* num_hits, uint32_t* counts,
                         uint32_t* ids, float* scores, int* status) {
    return guard([&] {
        if (!index || (n && (!requests || !num_hits || !counts || !ids || !scores))) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search_batch_flat: null argument");
        std::vector<const Request*> reqs(n);
        for (size_t i = 0; i < n; ++i) reqs[i] = requests[i] ? &requests[i]->req : nullptr;
        // Large batches run as a software pipeline of chunks over the index's two workspaces: while the GPU scans chunk c
        // the host compiles chunk c+1, and chunk c-1's merge + download run on the finish stream.
        bool any_deep = false;  // top + skip beyond one scan's ranking: such requests page on after their batch (not pipelined)
        for (size_t i = 0; i < n; ++i)
            if (reqs[i]) any_deep = any_deep || uint64_t(reqs[i]->top.value_or(10)) + uint64_t(reqs[i]->skip.value_or(0)) > uint64_t(vq::kMaxTopK);
        if (any_deep) {
            std::vector<std::unique_ptr<Result>> results;
            std::vector<int> st;
            std::vector<std::string> errs;
            {
                auto pb = run_partial(*index->idx, reqs.data(), n);
                finish_batch(*index->idx, *pb, nullptr, 1, results, st, errs);
            }
            complete_deep_requests(*index->idx, reqs.data(), n, results, st, errs);
            decline_explain(results, st, errs, "the flat batch path");
            copy_flat(results, st, errs, 0, stride, num_hits, counts, ids, scores, status);
            return;
        }
        static const size_t chunks_env = [] {
            const char* e = std::getenv("VQ_FLAT_CHUNKS");
            return size_t(e ? std::max(1, std::atoi(e)) : 0);
        }();
        // Requests with pre-passes (dictionary scans of fuzzy / prefix leaves, then unions) make the HOST side of a chunk long: it waits for
        // each pre-pass.  Two host threads, two chunks each, keep the GPU fed (config #4: 4.3 ms of kernels per chunk against 5 ms of
        // host-visible time; one thread: 50 k requests/s).  Plain batches stay on the calling thread: their host side is short.
        static const bool one_thread = std::getenv("VQ_FLAT_ONE_THREAD") != nullptr;
        static const size_t small_chunks = [] {  // a batch of 128 .. 511 requests with pre-passes: chunks of the two-thread pipeline (1: one piece)
            const char* e = std::getenv("VQ_FLAT_SMALL_CHUNKS");
            const int v = e ? std::atoi(e) : 2;
            return size_t(v == 4 ? 4 : v == 2 ? 2 : 1);
        }();
        const bool with_prepasses = kWorkspaces >= 4 && !one_thread && n >= 128 && batch_has_prepasses(reqs);
        // Plain batches on a large index: two chunks — the second is compiled while the first scans; launches of 256 requests fill the chip less
        // evenly than launches of 512 (100 M docs: 3-term AND 110.6 k requests/s with 2 chunks, 109.4 k with 1, 107.9 k with 4, 103.7 k with 8; the
        // request mix of config #5 68.5 k against 58.5 k).  On a small index the scan of a chunk is short next to its compilation and four chunks
        // hide more of the host side (10 M docs, AND + phrases + locality: 402 k with 4, 386 k with 2).
        const uint64_t shard_docs = uint64_t(index->idx->doc_hi) - index->idx->doc_lo;
        const size_t plain_chunks = shard_docs >= 32'000'000ull ? 2 : 4;
        const size_t nchunks = n >= 512 ? (chunks_env ? chunks_env : with_prepasses ? 4 : plain_chunks) : with_prepasses ? small_chunks : 1;
        std::vector<std::unique_ptr<PartialBatch>> inflight(nchunks);
        auto bounds = [&](size_t c) { return std::make_pair(n * c / nchunks, n * (c + 1) / nchunks); };
        auto finish = [&](size_t c) {
            std::vector<std::unique_ptr<Result>> results;
            std::vector<int> st;
            std::vector<std::string> errs;
            finish_batch(*index->idx, *inflight[c], nullptr, 1, results, st, errs);
            decline_explain(results, st, errs, "the flat batch path");
            copy_flat(results, st, errs, bounds(c).first, stride, num_hits, counts, ids, scores, status);
            inflight[c].reset();
        };
        const bool prepasses = with_prepasses && (nchunks == 4 || nchunks == 2);
        if (prepasses) {
            auto half = [&](size_t t) {
                for (size_t c = t; c < nchunks; c += 2) {
                    auto [b, e] = bounds(c);
                    inflight[c] = run_partial(*index->idx, reqs.data() + b, e - b, int(c));
                }
                for (size_t c = t; c < nchunks; c += 2) finish(c);
            };
            auto other = std::async(std::launch::async, half, size_t(1));
            try {
                half(0);
            } catch (...) {
                other.wait();
                throw;
            }
            other.get();
            return;
        }
        for (size_t c = 0; c < nchunks; ++c) {
            if (c >= size_t(kWorkspaces)) finish(c - kWorkspaces);
            auto [b, e] = bounds(c);
            inflight[c] = run_partial(*index->idx, reqs.data() + b, e - b, int(c % kWorkspaces));
        }
        for (size_t c = nchunks >= size_t(kWorkspaces) ? nchunks - kWorkspaces : 0; c < nchunks; ++c) finish(c);
    });
}

// ------------------------------------------------------------------ shard partials
int vq_search_batch_partial(const vq_index* index, const vq_request* const* requests, size_t n, vq_partial_batch** out) {
    return guard([&] {
        if (!index || !out || (n && !requests)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search_batch_partial: null argument");
        *out = nullptr;
        std::vector<const Request*> reqs(n);
        for (size_t i = 0; i < n; ++i) reqs[i] = requests[i] ? &requests[i]->req : nullptr;
        auto pb = run_partial(*index->idx, reqs.data(), n);
        // on the index's own stream the packed buffer must be complete before another library (RCCL) reads it; on a
        // caller-provided stream the caller's collective is ordered behind the scan by the stream itself
        if (index->idx->stream == index->idx->own_stream) VQ_HIP(hipStreamSynchronize(index->idx->stream));
        auto* h = new vq_partial_batch();
        h->pb = std::move(pb);
        *out = h;
    });
}
// A chunk of a sharded step whose chunks share ONE collective: workspace `slot` (0 .. VQ_PARTIAL_SLOTS-1, one per chunk in flight), partial placed at
// `arena_offset` (a multiple of 256) of the index's partial arena.
int vq_search_batch_partial_at(const vq_index* index, const vq_request* const* requests, size_t n, int slot, size_t arena_offset, vq_partial_batch** out) {
    return guard([&] {
        if (!index || !out || (n && !requests)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search_batch_partial_at: null argument");
        if (slot < 0 || slot >= vq::kWorkspaces) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_search_batch_partial_at: slot out of range");
        *out = nullptr;
        std::vector<const Request*> reqs(n);
        for (size_t i = 0; i < n; ++i) reqs[i] = requests[i] ? &requests[i]->req : nullptr;
        auto pb = run_partial(*index->idx, reqs.data(), n, slot, int64_t(arena_offset));
        if (index->idx->stream == index->idx->own_stream) VQ_HIP(hipStreamSynchronize(index->idx->stream));
        auto* h = new vq_partial_batch();
        h->pb = std::move(pb);
        *out = h;
    });
}
int vq_partial_slots(void) { return vq::kWorkspaces; }
void* vq_index_partial_arena_ptr(const vq_index* i) {
    if (!i) return nullptr;
    if (guard([&] {
            VQ_HIP(hipSetDevice(i->idx->device));
            i->idx->arena.ensure(vq::Index::kArenaBytes);
        }) != 0)
        return nullptr;
    return i->idx->arena.p;
}
size_t vq_partial_total_bytes(const vq_partial_batch* p) { return p ? size_t(p->pb->layout.bytes) : 0; }
size_t vq_partial_bytes(const vq_partial_batch* p) { return p ? size_t(p->pb->layout.off_hist) : 0; }
void* vq_partial_device_ptr(vq_partial_batch* p) { return p ? p->pb->d_partial : nullptr; }
size_t vq_partial_hist_bytes(const vq_partial_batch* p) { return p ? size_t(p->pb->layout.total_hist) * 4 : 0; }
void* vq_partial_hist_device_ptr(vq_partial_batch* p) { return p && p->pb->d_partial ? p->pb->d_partial + p->pb->layout.off_hist : nullptr; }

// the flat merges rank top + skip <= kMaxTopK per request (vq_merge_partials hands page 0 back instead: vq_result_is_page / vq_request_page_after)
static void decline_deep(std::vector<std::unique_ptr<Result>>& results, std::vector<int>& st, std::vector<std::string>& errs) {
    for (size_t i = 0; i < results.size(); ++i)
        if (st[i] == 0 && results[i] && results[i]->deep) {
            st[i] = VQ_ERR_UNSUPPORTED;
            errs[i] = "unsupported on the MI355X query path: top + skip > " + std::to_string(vq::kMaxTopK) + " on the sharded partial / merge path";
            results[i].reset();
        }
}

int vq_merge_partials(const vq_index* index, vq_partial_batch* local, const void* gathered_device, uint32_t num_shards, vq_result** out, int* status) {
    return guard([&] {
        if (!index || !local || !out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_merge_partials: null argument");
        std::vector<std::unique_ptr<Result>> results;
        std::vector<int> st;
        std::vector<std::string> errs;
        finish_batch(*index->idx, *local->pb, gathered_device, num_shards, results, st, errs);
        // (a request that reaches beyond one scan's ranking comes back as its page 0, vq_result_is_page: the caller pages on, on all shards)
        decline_explain(results, st, errs, "the sharded partial / merge path");
        for (size_t i = 0; i < results.size(); ++i) {
            out[i] = nullptr;
            if (status) status[i] = st[i];
            if (st[i] == 0) {
                auto* r = new vq_result();
                r->r = std::move(*results[i]);
                out[i] = r;
            } else if (g_err.empty()) g_err = errs[i];
        }
    });
}
int vq_merge_partials_flat(const vq_index* index, vq_partial_batch* local, const void* gathered_device, uint32_t num_shards, size_t stride,
                           uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status) {
    return guard([&] {
        if (!index || !local || !num_hits || !counts || !ids || !scores) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_merge_partials_flat: null argument");
        std::vector<std::unique_ptr<Result>> results;
        std::vector<int> st;
        std::vector<std::string> errs;
        finish_batch(*index->idx, *local->pb, gathered_device, num_shards, results, st, errs);
        decline_deep(results, st, errs);
        decline_explain(results, st, errs, "the sharded partial / merge path");
        copy_flat(results, st, errs, 0, stride, num_hits, counts, ids, scores, status);
    });
}
int vq_merge_partials_flat_strided(const vq_index* index, vq_partial_batch* local, const void* gathered_device, uint32_t num_shards, size_t shard_stride,
                                   size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status) {
    return guard([&] {
        if (!index || !local || !num_hits || !counts || !ids || !scores) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_merge_partials_flat_strided: null argument");
        std::vector<std::unique_ptr<Result>> results;
        std::vector<int> st;
        std::vector<std::string> errs;
        finish_batch(*index->idx, *local->pb, gathered_device, num_shards, results, st, errs, shard_stride);
        decline_deep(results, st, errs);
        decline_explain(results, st, errs, "the sharded partial / merge path");
        copy_flat(results, st, errs, 0, stride, num_hits, counts, ids, scores, status);
    });
}
void vq_partial_free(vq_partial_batch* p) { delete p; }

// ------------------------------------------------------------------ the sharded step inside the library
// compile -> scans (scan stream) -> all-gather of the packed partials + all-reduce of the facet histograms (RCCL, collective stream) ->
// merge + download (finish stream): one call per step, no interpreter and no tensor library between the stages.  RCCL is loaded at run time
// (librccl.so.1 — the copy already in the process when the caller's framework brought one).
namespace {
typedef struct ncclComm* ncclComm_t;
struct NcclId {
    char internal[VQ_COMM_ID_BYTES];
};
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;                 // optional
    int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;   // optional
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint8 = 1, kNcclUint32 = 3, kNcclUint64 = 5, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t (rccl.h)
static Rccl& rccl_api() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.h) break;
        }
        if (!x.h) return x;
        auto sym = [&](const char* n) { return dlsym(x.h, n); };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
        x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(sym("ncclCommAbort"));
        x.CommGetAsyncError = reinterpret_cast<decltype(x.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        return x;
    }();
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce)
        throw VelociError(VQ_ERR_DEVICE, "RCCL (librccl.so.1) is not available: the in-library exchange of a sharded step needs it");
    return r;
}
// RCCL greets on stdout when it starts up: a host program's stdout is not this library's to write to (the greeting goes to stderr)
int quiet_stdout(const std::function<int()>& f) {
    std::fflush(stdout);
    const int keep = dup(1);
    if (keep >= 0) (void)dup2(2, 1);
    const int rc = f();
    std::fflush(stdout);
    if (keep >= 0) {
        (void)dup2(keep, 1);
        (void)close(keep);
    }
    return rc;
}
void nccl_check(int rc, const char* what) {
    if (rc != 0) throw VelociError(VQ_ERR_DEVICE, std::string(what) + ": " + (rccl_api().GetErrorString ? rccl_api().GetErrorString(rc) : "RCCL error"));
}
// sums over the shards before compilation (result sizes of count pre-passes, merged list lengths): the in-library form of vq_index_set_allreduce
int comm_sum_u64(void* ctx, uint64_t* values, size_t n) {
    const Index& idx = *static_cast<const Index*>(ctx);
    ShardComm& c = *idx.comm;
    std::lock_guard<std::mutex> g(c.mu);
    try {
        VQ_HIP(hipSetDevice(idx.device));
        c.red.ensure(n * 8);
        VQ_HIP(hipMemcpyAsync(c.red.p, values, n * 8, hipMemcpyHostToDevice, c.stream));
        nccl_check(rccl_api().AllReduce(c.red.p, c.red.p, n, kNcclUint64, kNcclSum, static_cast<ncclComm_t>(c.nccl), c.stream), "ncclAllReduce");
        VQ_HIP(hipMemcpyAsync(values, c.red.p, n * 8, hipMemcpyDeviceToHost, c.stream));
        VQ_HIP(hipStreamSynchronize(c.stream));
        return 0;
    } catch (...) {
        return -1;
    }
}
void comm_setup(Index& idx, std::unique_ptr<ShardComm> c) {
    VQ_HIP(hipSetDevice(idx.device));
    {
        int lo = 0, hi = 0;
        VQ_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        VQ_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));  // (collectives are short and latency-bound: ahead of the next step's scans)
    }
    for (int k = 0; k < 2; ++k) {
        VQ_HIP(hipEventCreateWithFlags(&c->ev_scan[k], hipEventDisableTiming));
        VQ_HIP(hipEventCreateWithFlags(&c->ev_xchg[k], hipEventDisableTiming));
        VQ_HIP(hipEventCreateWithFlags(&c->ev_fin[k], hipEventDisableTiming));
    }
    idx.comm = std::move(c);
}
}  // namespace

vq::ShardComm::~ShardComm() {
    if (nccl) (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(nccl));
    for (int k = 0; k < 2; ++k) {
        if (ev_scan[k]) (void)hipEventDestroy(ev_scan[k]);
        if (ev_xchg[k]) (void)hipEventDestroy(ev_xchg[k]);
        if (ev_fin[k]) (void)hipEventDestroy(ev_fin[k]);
    }
    if (stream) (void)hipStreamDestroy(stream);
}

// The index lets go of its communicator — refused while a step made with it is still alive (the step's destructor and its queued merge
// refer to it); the sum hook that points into it goes first, so that a failed re-initialisation leaves no hook without a communicator.
static void comm_release(Index& idx, const char* who) {
    if (idx.comm && idx.comm->live != 0)
        throw VelociError(VQ_ERR_INVALID_ARGUMENT, std::string(who) + ": " + std::to_string(idx.comm->live) + " step(s) of this index are still in flight (vq_shard_step_end / vq_shard_step_free them first)");
    if (idx.allreduce_fn == comm_sum_u64) {
        idx.allreduce_fn = nullptr;
        idx.allreduce_ctx = nullptr;
    }
    idx.comm.reset();
}

int vq_comm_unique_id(void* id_out) {
    return guard([&] {
        if (!id_out) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_comm_unique_id: null argument");
        NcclId id;
        nccl_check(quiet_stdout([&] { return rccl_api().GetUniqueId(&id); }), "ncclGetUniqueId");
        std::memcpy(id_out, &id, sizeof id);
    });
}
int vq_comm_init(vq_index* index, int nranks, int rank, const void* unique_id) {
    return guard([&] {
        if (!index || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_comm_init: bad argument");
        Index& idx = *index->idx;
        comm_release(idx, "vq_comm_init");
        VQ_HIP(hipSetDevice(idx.device));
        auto c = std::make_unique<ShardComm>();
        c->nranks = nranks;
        c->rank = rank;
        NcclId id;
        std::memcpy(&id, unique_id, sizeof id);
        ncclComm_t comm = nullptr;
        nccl_check(quiet_stdout([&] { return rccl_api().CommInitRank(&comm, nranks, id, rank); }), "ncclCommInitRank");
        c->nccl = comm;
        comm_setup(idx, std::move(c));
        idx.allreduce_fn = comm_sum_u64;
        idx.allreduce_ctx = &idx;
    });
}
int vq_comm_init_custom(vq_index* index, int nranks, int rank, vq_allgather_fn allgather, vq_allreduce_u32_fn allreduce_u32, void* ctx) {
    return guard([&] {
        if (!index || !allgather || nranks < 1 || rank < 0 || rank >= nranks) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_comm_init_custom: bad argument");
        auto c = std::make_unique<ShardComm>();
        c->nranks = nranks;
        c->rank = rank;
        c->allgather = allgather;
        c->allreduce_u32 = allreduce_u32;
        c->ctx = ctx;
        comm_release(*index->idx, "vq_comm_init_custom");
        comm_setup(*index->idx, std::move(c));
    });
}
int vq_comm_destroy(vq_index* index) {
    return guard([&] {
        if (!index) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_comm_destroy: null index");
        comm_release(*index->idx, "vq_comm_destroy");
    });
}

struct vq_shard_step {
    const Index* idx = nullptr;
    std::vector<std::unique_ptr<PartialBatch>> pbs;
    std::vector<size_t> first, arena_off;  // chunk c: its first request, the offset of its gathered copies inside the step's gather buffer
    size_t n = 0, used = 0;
    int parity = 0;
    bool merge_queued = false;
    bool counted = false;
    ~vq_shard_step() {
        pbs.clear();  // (the workspaces first: a parity is free once nothing of the step holds them)
        if (idx && idx->comm && idx->comm->unmerged == this) idx->comm->unmerged = nullptr;
        if (counted && idx && idx->comm) {
            idx->comm->live -= 1;
            idx->comm->busy[parity] = false;
        }
    }
};
namespace {
// A step failed as a whole on this rank (an exception between its collectives: out of device memory in a pre-pass, a failed launch), or did not
// complete in time: this rank's exchanges no longer line up with the other ranks'.  The communicator is marked down — every later step on it is
// refused — and an RCCL communicator is aborted, so that ranks waiting for this one inside a collective see an error instead of waiting for ever.
void comm_fail(ShardComm& c, const std::string& why) {
    if (!c.failed.empty()) return;
    c.failed = why;
    if (c.nccl) {
        try {
            if (rccl_api().CommAbort) (void)rccl_api().CommAbort(static_cast<ncclComm_t>(c.nccl));
            c.nccl = nullptr;  // (aborted: nothing is left to destroy)
        } catch (...) {
        }
    }
}
// Wait for a step's last event, but not for ever: a rank that never joined an exchange (it failed before, or died) must end this rank's step with
// an error, not hang it.  VQ_STEP_TIMEOUT_MS (default 120 000; 0: no limit).  An error RCCL reports asynchronously (a peer gone) ends the wait at once.
void bounded_wait(const Index& idx, ShardComm& c, hipEvent_t ev) {
    static const long limit_ms = [] {
        const char* e = std::getenv("VQ_STEP_TIMEOUT_MS");
        return e ? std::atol(e) : 120000L;
    }();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) {
            comm_fail(c, std::string("the step's stream reported ") + hipGetErrorString(q));
            throw VelociError(VQ_ERR_DEVICE, "sharded step: " + c.failed);
        }
        if (spins < 2000) continue;  // (the usual case: the step is about to finish)
        if ((spins & 63u) == 0) {
            if (c.nccl && rccl_api().CommGetAsyncError) {
                int aerr = 0;
                if (rccl_api().CommGetAsyncError(static_cast<ncclComm_t>(c.nccl), &aerr) == 0 && aerr != 0) {
                    comm_fail(c, std::string("RCCL reported ") + (rccl_api().GetErrorString ? rccl_api().GetErrorString(aerr) : "an error") + " while the step was in flight (a rank is gone)");
                    throw VelociError(VQ_ERR_DEVICE, "sharded step: " + c.failed);
                }
            }
            const long waited = long(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count());
            if (limit_ms > 0 && waited > limit_ms) {
                comm_fail(c, "a step did not complete within " + std::to_string(limit_ms) + " ms (VQ_STEP_TIMEOUT_MS): a rank did not join its exchange");
                throw VelociError(VQ_ERR_DEVICE, "sharded step: " + c.failed);
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    (void)idx;
}
// The step's merge and download go onto the finish stream, behind its exchange.  The NEXT step's scans are ordered behind them (ev_fin): a
// scan launch fills every wave slot of the chip at once, and a merge queued beside it would wait for slots to free up (measured: 0.8-1.5 ms
// for a kernel that takes 8 us on an idle chip) — the GPU idles for the tens of microseconds of the exchange instead.
void step_queue_merge(vq_shard_step& step) {
    if (step.merge_queued) return;
    step.merge_queued = true;
    const Index& idx = *step.idx;
    ShardComm& c = *idx.comm;
    if (c.unmerged == &step) c.unmerged = nullptr;
    if (step.used) VQ_HIP(hipStreamWaitEvent(idx.fin_stream, c.ev_xchg[step.parity], 0));
    const uint8_t* gathered = static_cast<const uint8_t*>(c.gathered[step.parity].p);
    for (size_t k = 0; k < step.pbs.size(); ++k)
        finish_launch(idx, *step.pbs[k], step.used && step.pbs[k]->nq_dev ? gathered + step.arena_off[k] : nullptr, uint32_t(c.nranks));
    VQ_HIP(hipEventRecord(c.ev_fin[step.parity], idx.fin_stream));
}
}  // namespace

int vq_shard_step_begin(const vq_index* index, const vq_request* const* requests, size_t n, vq_shard_step** out) {
    return guard([&] {
        if (!index || !out || (n && !requests)) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_shard_step_begin: null argument");
        *out = nullptr;
        const Index& idx = *index->idx;
        if (!idx.comm) {  // no communicator: the index is the only shard — the same pipeline without an exchange (two steps in flight on one GPU)
            auto local = std::make_unique<ShardComm>();
            comm_setup(const_cast<Index&>(idx), std::move(local));
        }
        ShardComm& c = *idx.comm;
        if (!c.failed.empty()) throw VelociError(VQ_ERR_DEVICE, "vq_shard_step_begin: the communicator is down (" + c.failed + "): make it again (vq_comm_init / vq_comm_init_custom) on every rank");
        const bool exchange = c.nccl != nullptr || c.allgather != nullptr;
        static const bool timing = std::getenv("VQ_TIMING") != nullptr;
        const auto tb0 = std::chrono::steady_clock::now();
        VQ_HIP(hipSetDevice(idx.device));
        std::vector<const Request*> reqs(n);
        for (size_t i = 0; i < n; ++i) reqs[i] = requests[i] ? &requests[i]->req : nullptr;
        if (c.unmerged) {  // the step before this one: its merge goes out first, this step's scans run behind it
            vq_shard_step& prev = *static_cast<vq_shard_step*>(c.unmerged);
            step_queue_merge(prev);
            VQ_HIP(hipStreamWaitEvent(idx.stream, c.ev_fin[prev.parity], 0));
        }
        const auto tb1 = std::chrono::steady_clock::now();
        if (c.live >= 2) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_shard_step_begin: two steps are in flight already (end one first)");
        auto step = std::make_unique<vq_shard_step>();
        step->idx = &idx;
        step->counted = true;
        c.live += 1;
        step->n = n;
        // the parity (workspace pair, gather buffer, events) no live step holds — not a running count: steps may end in any order, and a
        // begin that throws (a pre-pass out of memory) is retried while the other step is still in flight
        step->parity = c.busy[0] ? 1 : 0;
        if (c.busy[step->parity]) {
            step->counted = false;
            c.live -= 1;
            throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_shard_step_begin: both step slots are held (end a step first)");
        }
        c.busy[step->parity] = true;
        auto tb2 = tb1;
        try {
        // One launch per step: with two steps in flight the compilation of step i + 1 already overlaps the scans of step i, and one launch of
        // 1024 requests fills the chip more evenly than two of 512 (100 M docs: 7.25 against 7.41 ms per step).  VQ_SHARD_CHUNKS=2 cuts a step in two.
        // (read at every step: a host program may cut only some of its steps in two — bench.py does for the headline's launches of distinct queries)
        const size_t chunks_env = [] {
            const char* e = std::getenv("VQ_SHARD_CHUNKS");
            return size_t(e ? std::min(2, std::max(1, std::atoi(e))) : 0);
        }();
        const size_t nchunks = chunks_env && n >= 512 ? chunks_env : 1;
        for (size_t k = 0; k < nchunks; ++k) {
            const size_t b = n * k / nchunks, e = n * (k + 1) / nchunks;
            step->first.push_back(b);
            step->pbs.push_back(run_partial(idx, reqs.data() + b, e - b, step->parity * 2 + int(k)));
        }
        tb2 = std::chrono::steady_clock::now();
        // ---- the exchange: behind the step's scans, on the collective stream.  Per chunk the part of its partial in front of the histograms
        // (hit counts, statistics, top-k keys: the same size on every rank) is all-gathered, the histograms are summed in place.
        size_t off = 0;
        for (auto& pb : step->pbs) {
            step->arena_off.push_back(off);
            if (exchange && pb->nq_dev) off += (size_t(pb->layout.off_hist) * size_t(c.nranks) + 255) / 256 * 256;
        }
        step->used = off;
        if (off) {
            VQ_HIP(hipEventRecord(c.ev_scan[step->parity], idx.stream));
            VQ_HIP(hipStreamWaitEvent(c.stream, c.ev_scan[step->parity], 0));
            c.gathered[step->parity].ensure(off);
            uint8_t* gathered = c.gathered[step->parity].as<uint8_t>();
            for (size_t k = 0; k < step->pbs.size(); ++k) {
                PartialBatch& pb = *step->pbs[k];
                if (!pb.nq_dev) continue;
                const size_t bytes = size_t(pb.layout.off_hist);
                if (c.nccl) nccl_check(rccl_api().AllGather(pb.d_partial, gathered + step->arena_off[k], bytes, kNcclUint8, static_cast<ncclComm_t>(c.nccl), c.stream), "ncclAllGather");
                else if (c.allgather(c.ctx, pb.d_partial, gathered + step->arena_off[k], bytes, c.stream) != 0) throw VelociError(VQ_ERR_DEVICE, "sharded step: the caller's all-gather failed");
                if (pb.layout.total_hist) {  // facet counts are additive: summed in place, read by this rank's merge
                    void* h = pb.d_partial + pb.layout.off_hist;
                    if (c.nccl) nccl_check(rccl_api().AllReduce(h, h, size_t(pb.layout.total_hist), kNcclUint32, kNcclSum, static_cast<ncclComm_t>(c.nccl), c.stream), "ncclAllReduce");
                    else if (!c.allreduce_u32 || c.allreduce_u32(c.ctx, h, size_t(pb.layout.total_hist), c.stream) != 0)
                        throw VelociError(VQ_ERR_DEVICE, "sharded step: the caller's all-reduce failed");
                }
            }
            VQ_HIP(hipEventRecord(c.ev_xchg[step->parity], c.stream));
        }
        } catch (const std::exception& ex) {
            // Whatever was thrown from here on ends the step on THIS rank only (requests that fail on their own ride along as statuses and never
            // throw): with other ranks around, their exchange of this step will miss this rank — the communicator goes down with the step.
            if (exchange && c.nranks > 1) comm_fail(c, std::string("a step failed on rank ") + std::to_string(c.rank) + ": " + ex.what());
            throw;
        }
        c.unmerged = step.get();
        *out = step.release();
        if (timing) {
            const auto tb3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[vq timing] shard step begin %.3f ms (queue the previous merge %.3f, compile + scans %.3f, exchange calls %.3f)\n", ms(tb0, tb3), ms(tb0, tb1), ms(tb1, tb2), ms(tb2, tb3));
        }
    });
}

int vq_shard_step_end(vq_shard_step* step_raw, size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids, float* scores, int* status) {
    std::unique_ptr<vq_shard_step> step(step_raw);  // freed whatever happens
    return guard([&] {
        if (!step || (step->n && (!num_hits || !counts || !ids || !scores))) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_shard_step_end: null argument");
        const Index& idx = *step->idx;
        ShardComm& c = *idx.comm;
        VQ_HIP(hipSetDevice(idx.device));
        static const bool timing = std::getenv("VQ_TIMING") != nullptr;
        const auto te0 = std::chrono::steady_clock::now();
        if (!c.failed.empty()) throw VelociError(VQ_ERR_DEVICE, "vq_shard_step_end: the communicator is down (" + c.failed + ")");
        step_queue_merge(*step);
        bounded_wait(idx, c, c.ev_fin[step->parity]);
        if (timing) {
            VQ_HIP(hipStreamSynchronize(idx.fin_stream));
            std::fprintf(stderr, "[vq timing] shard step end: waited %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count());
        }
        const uint8_t* gathered = static_cast<const uint8_t*>(c.gathered[step->parity].p);
        for (size_t k = 0; k < step->pbs.size(); ++k) {
            std::vector<std::unique_ptr<Result>> results;
            std::vector<int> st;
            std::vector<std::string> errs;
            finish_batch(idx, *step->pbs[k], step->used && step->pbs[k]->nq_dev ? gathered + step->arena_off[k] : nullptr, uint32_t(c.nranks), results, st, errs);
            // a request that reaches beyond one scan's ranking: on an index that answers alone (no exchange) its further pages are scanned here, like on
            // the flat batch path; over shards every rank would have to page on the MERGED page — declined there (vq_merge_partials pages)
            if (!step->used && c.nranks == 1) complete_deep_requests(idx, step->pbs[k]->reqs.data(), step->pbs[k]->reqs.size(), results, st, errs);
            else decline_deep(results, st, errs);
            decline_explain(results, st, errs, "the sharded step");
            copy_flat(results, st, errs, step->first[k], stride, num_hits, counts, ids, scores, status);
            step->pbs[k].reset();  // (destroyed here, by the thread that uses these allocations next: handing the frees to a background thread was tried and
                                   //  cost more than it saved — it frees into the arenas the compile threads are allocating from)
        }
        if (timing) std::fprintf(stderr, "[vq timing] shard step end total %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count());
    });
}
void vq_shard_step_free(vq_shard_step* step) { delete step; }

int vq_shard_step_flat(const vq_index* index, const vq_request* const* requests, size_t n, size_t stride, uint64_t* num_hits, uint32_t* counts, uint32_t* ids,
                       float* scores, int* status) {
    vq_shard_step* step = nullptr;
    const int rc = vq_shard_step_begin(index, requests, n, &step);
    if (rc != VQ_OK) return rc;
    return vq_shard_step_end(step, stride, num_hits, counts, ids, scores, status);
}

// ------------------------------------------------------------------ measurement
int vq_profile_enable(vq_index* i, int on) {
    return guard([&] {
        if (!i) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_profile_enable: null index");
        std::lock_guard<std::mutex> g(i->idx->profile_mutex);
        i->idx->profile.enabled = on != 0;
    });
}
static bool is_scan_kernel(int k) { return k >= K_SCAN_LEAF_F32 && k <= K_TILE_SCAN; }
int vq_profile_read(const vq_index* i, int reset, double* scan_kernel_ms, uint64_t* scan_launches, uint64_t* algorithmic_bytes) {
    return guard([&] {
        if (!i) throw VelociError(VQ_ERR_INVALID_ARGUMENT, "vq_profile_read: null index");
        std::lock_guard<std::mutex> g(i->idx->profile_mutex);
        Profile& p = i->idx->profile;
        double ms = 0;
        uint64_t algo = 0;
        for (int k = 0; k < K_COUNT_; ++k)
            if (is_scan_kernel(k)) {
                ms += p.k[k].ms;
                algo += p.k[k].algorithmic_bytes;
            }
        if (scan_kernel_ms) *scan_kernel_ms = ms;
        if (scan_launches) *scan_launches = p.batches;
        if (algorithmic_bytes) *algorithmic_bytes = algo;
        if (reset) p = Profile{p.enabled};
    });
}
const char* vq_profile_json(const vq_index* i, int reset) {
    thread_local std::string out;
    out = "{}";
    if (!i) return out.c_str();
    std::lock_guard<std::mutex> g(i->idx->profile_mutex);
    Profile& p = i->idx->profile;
    out = "{\"batches\":" + std::to_string(p.batches) + ",\"kernels\":{";
    bool first = true;
    for (int k = 0; k < K_COUNT_; ++k) {
        const KernelProfile& kp = p.k[k];
        if (!kp.launches) continue;
        char buf[512];
        std::snprintf(buf, sizeof buf, "%s\"%s\":{\"ms\":%.6f,\"launches\":%llu,\"layout_bytes\":%llu,\"algorithmic_bytes\":%llu,\"gathered_bytes\":%llu,\"queries\":%llu,\"scan\":%s}",
                      first ? "" : ",", kKernelNames[k], kp.ms, (unsigned long long)kp.launches, (unsigned long long)kp.layout_bytes,
                      (unsigned long long)kp.algorithmic_bytes, (unsigned long long)kp.gathered_bytes, (unsigned long long)kp.queries, is_scan_kernel(k) ? "true" : "false");
        out += buf;
        first = false;
    }
    out += "}}";
    if (reset) p = Profile{p.enabled};
    return out.c_str();
}

#ifdef VQ_STAMP
void vq_debug_stamps(unsigned long long* out, int reset) { vq::debug_read_stamps(out, reset); }
void vq_debug_probe_stamps(unsigned long long* out, int reset) { vq::debug_read_probe_stamps(out, reset); }
void vq_debug_ring_stamps(unsigned long long* out, int reset) { vq::debug_read_ring_stamps(out, reset); }
#endif

}  // extern "C"
