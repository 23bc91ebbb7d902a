// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  C entry points (ctypes) for the CPU oracle.
// Index loading mirrors include/veloci_amd.h one to one (vo_ instead of vq_) so tests feed the
// oracle and the HIP library from the same arrays.
#include <malloc.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "veloci_oracle.hpp"

using namespace vo;

namespace {
thread_local std::string g_err;
struct ResultBox {
    SearchResult r;
    std::vector<uint32_t> ids;
    std::vector<float> scores;
    std::string why_json, why_info_json, explain_json;
};
int fail(const VelociError& e) {
    g_err = e.what();
    return e.code;
}
}  // namespace

extern "C" {

const char* vo_last_error() { return g_err.c_str(); }

void* vo_index_new(uint32_t num_anchors) {
    auto* i = new Index();
    i->num_anchors = num_anchors;
    return i;
}
void vo_index_free(void* p) { delete static_cast<Index*>(p); }

int vo_index_add_fst(void* p, const char* path, uint32_t num_terms, const uint8_t* bytes, const uint64_t* offsets) {
    Fst f;
    f.terms.reserve(num_terms);
    for (uint32_t i = 0; i < num_terms; ++i) f.terms.emplace_back(reinterpret_cast<const char*>(bytes) + offsets[i], size_t(offsets[i + 1] - offsets[i]));
    f.finalize();
    static_cast<Index*>(p)->fst[path] = std::move(f);
    return 0;
}
int vo_index_add_token_to_anchor_score(void* p, const char* path, uint32_t num_tokens, const uint64_t* offsets, const uint32_t* anchors,
                                       const uint32_t* scores, const uint64_t* /*global_lens*/) {
    TokenToAnchorScore t;
    t.offsets.assign(offsets, offsets + num_tokens + 1);
    uint64_t n = offsets[num_tokens];
    t.anchors.assign(anchors, anchors + n);
    t.scores_f16.resize(n);
    for (uint64_t i = 0; i < n; ++i) t.scores_f16[i] = f32_to_f16_bits(float(scores[i]));  // token_to_anchor_score_vint.rs:155
    static_cast<Index*>(p)->token_to_anchor_score[path] = std::move(t);
    return 0;
}
int vo_index_add_key_value_store(void* p, const char* path, uint32_t key_base, uint32_t num_keys, const uint64_t* offsets, const uint32_t* values) {
    KeyValueStore kv;
    kv.key_base = key_base;
    kv.offsets.assign(offsets, offsets + num_keys + 1);
    kv.values.assign(values, values + offsets[num_keys]);
    static_cast<Index*>(p)->key_value_stores[path] = std::move(kv);
    return 0;
}
int vo_index_add_phrase_pair_to_anchor(void* p, const char* path, uint64_t num_pairs, const uint32_t* t1, const uint32_t* t2, const uint64_t* offsets,
                                       const uint32_t* anchors) {
    PhrasePairToAnchor pp;
    pp.keys.reserve(num_pairs);
    for (uint64_t i = 0; i < num_pairs; ++i) pp.keys.push_back({t1[i], t2[i]});
    pp.offsets.assign(offsets, offsets + num_pairs + 1);
    pp.anchors.assign(anchors, anchors + offsets[num_pairs]);
    static_cast<Index*>(p)->phrase_pair_to_anchor[path] = std::move(pp);
    return 0;
}
int vo_index_add_boost(void* p, const char* path, uint32_t key_base, uint32_t num_keys, const uint8_t* present, const uint32_t* bits) {
    BoostStore b;
    b.key_base = key_base;
    if (present) b.present.assign(present, present + num_keys);
    b.bits.assign(bits, bits + num_keys);
    static_cast<Index*>(p)->boost_valueid_to_value[path] = std::move(b);
    return 0;
}
int vo_index_set_column_meta(void* p, const char* field, int is_anchor_identity_column, int tokenize) {
    ColumnMeta m;
    m.is_anchor_identity_column = is_anchor_identity_column != 0;
    m.tokenize = tokenize != 0;
    static_cast<Index*>(p)->columns[field] = m;
    return 0;
}

// ---- search
int vo_search_json(const void* index, const char* json, size_t len, void** out) {
    try {
        Request req = request_from_json_text(json, len);
        auto* box = new ResultBox();
        box->r = search(std::move(req), *static_cast<const Index*>(index));
        for (auto& h : box->r.data) {
            box->ids.push_back(h.id);
            box->scores.push_back(h.score);
        }
        *out = box;
        return 0;
    } catch (const VelociError& e) {
        *out = nullptr;
        return fail(e);
    }
}
// why_found_terms (search.rs:186) as JSON {"<path>": ["term", ...]}
const char* vo_result_why_found_terms_json(const void* r) {
    auto* box = const_cast<ResultBox*>(static_cast<const ResultBox*>(r));
    std::string& s = box->why_json;
    s = "{";
    bool first = true;
    for (auto& kv : box->r.why_found_terms) {
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
    return s.c_str();
}
// why_found_info (search.rs:220-224) as JSON {"<anchor id>": {"<field>": ["highlighted text", ...]}}
const char* vo_result_why_found_info_json(const void* r) {
    auto* box = const_cast<ResultBox*>(static_cast<const ResultBox*>(r));
    std::string& s = box->why_info_json;
    s = "{";
    bool first = true;
    for (auto& [anchor, fields] : box->r.why_found_info) {
        if (!first) s += ',';
        first = false;
        s += '"' + std::to_string(anchor) + "\":{";
        bool f1 = true;
        for (auto& [field, texts] : fields) {
            if (!f1) s += ',';
            f1 = false;
            vqjson::escape_to(s, field);
            s += ":[";
            for (size_t i = 0; i < texts.size(); ++i) {
                if (i) s += ',';
                vqjson::escape_to(s, texts[i]);
            }
            s += ']';
        }
        s += '}';
    }
    s += '}';
    return s.c_str();
}
// explain (search.rs:86,96: result.explain.get(&hit.id)) as a JSON array parallel to the hits: null or the hit's records
const char* vo_result_explain_json(const void* r) {
    auto* box = const_cast<ResultBox*>(static_cast<const ResultBox*>(r));
    std::string& s = box->explain_json;
    s = "[";
    for (size_t i = 0; i < box->r.data.size(); ++i) {
        if (i) s += ',';
        auto it = box->r.explain.find(box->r.data[i].id);
        s += it == box->r.explain.end() ? std::string("null") : explain_json(it->second);
    }
    s += ']';
    return s.c_str();
}
// suggest (search_field.rs:194-231): `json` is a Request with "suggest" parts, or a bare RequestSearchPart (then top / skip are the part's)
struct SuggestBox {
    std::vector<SuggestEntry> e;
};
int vo_suggest_json(const void* index, const char* json, size_t len, void** out) {
    try {
        *out = nullptr;
        vqjson::Value v = vqjson::parse(json, len);
        Request req;
        if (v.is_object() && v.get("suggest")) req = request_from_json(v);
        else {
            RequestSearchPart part = search_part_from_json(v);
            req.suggest = std::vector<RequestSearchPart>{part};
            req.top = part.top;  // :226-227
            req.skip = part.skip;
        }
        auto* box = new SuggestBox();
        box->e = suggest_multi(*static_cast<const Index*>(index), std::move(req));
        *out = box;
        return 0;
    } catch (const VelociError& e) {
        return fail(e);
    } catch (const vqjson::ParseError& e) {
        return fail(VelociError(ERR_JSON, std::string("JsonError: ") + e.what()));
    }
}
// highlight (search_field.rs:233-245): `json` is a bare RequestSearchPart; the result is read with the vo_suggest_* accessors (text = snippet)
int vo_highlight_json(const void* index, const char* json, size_t len, void** out) {
    try {
        *out = nullptr;
        RequestSearchPart part = search_part_from_json(vqjson::parse(json, len));
        auto* box = new SuggestBox();
        try {
            box->e = highlight(*static_cast<const Index*>(index), std::move(part));
        } catch (...) {
            delete box;
            throw;
        }
        *out = box;
        return 0;
    } catch (const VelociError& e) {
        return fail(e);
    } catch (const vqjson::ParseError& e) {
        return fail(VelociError(ERR_JSON, std::string("JsonError: ") + e.what()));
    }
}
// highlight_text (highlight_field.rs:92-146): terms as a JSON array of strings, snippet_info as JSON or empty.  Returns a malloc'ed C string (vo_string_free),
// NULL with *none = 1 when nothing is highlighted, NULL with *none = 0 on an error (vo_last_error)
char* vo_highlight_text(const char* text, size_t len, const char* terms_json, size_t terms_len, const char* snippet_info_json, size_t si_len, int tokenized, int* none) {
    *none = 0;
    try {
        std::set<std::string> set;
        vqjson::Value tv = vqjson::parse(terms_json, terms_len);
        for (auto& e : tv.arr) set.insert(e.str);
        SnippetInfo opt;
        if (si_len) opt = snippet_info_from_json(vqjson::parse(snippet_info_json, si_len));
        auto r = highlight_text(std::string(text, len), set, opt, tokenized != 0);
        if (!r) {
            *none = 1;
            return nullptr;
        }
        char* out = static_cast<char*>(std::malloc(r->size() + 1));
        std::memcpy(out, r->data(), r->size());
        out[r->size()] = 0;
        return out;
    } catch (const VelociError& e) {
        fail(e);
        return nullptr;
    } catch (const vqjson::ParseError& e) {
        fail(VelociError(ERR_JSON, std::string("JsonError: ") + e.what()));
        return nullptr;
    }
}
void vo_string_free(char* s) { std::free(s); }
size_t vo_suggest_len(const void* s) { return static_cast<const SuggestBox*>(s)->e.size(); }
const char* vo_suggest_text(const void* s, size_t i) { return static_cast<const SuggestBox*>(s)->e[i].text.c_str(); }
float vo_suggest_score(const void* s, size_t i) { return static_cast<const SuggestBox*>(s)->e[i].score; }
uint32_t vo_suggest_term_id(const void* s, size_t i) { return static_cast<const SuggestBox*>(s)->e[i].term_id; }
void vo_suggest_free(void* s) { delete static_cast<SuggestBox*>(s); }

uint64_t vo_result_num_hits(const void* r) { return static_cast<const ResultBox*>(r)->r.num_hits; }
uint64_t vo_result_execution_time_ns(const void* r) { return static_cast<const ResultBox*>(r)->r.execution_time_ns; }
size_t vo_result_len(const void* r) { return static_cast<const ResultBox*>(r)->ids.size(); }
const uint32_t* vo_result_ids(const void* r) { return static_cast<const ResultBox*>(r)->ids.data(); }
const float* vo_result_scores(const void* r) { return static_cast<const ResultBox*>(r)->scores.data(); }
size_t vo_result_num_facets(const void* r) { return static_cast<const ResultBox*>(r)->r.facets.size(); }
const char* vo_result_facet_field(const void* r, size_t f) { return static_cast<const ResultBox*>(r)->r.facets[f].first.c_str(); }
size_t vo_result_facet_len(const void* r, size_t f) { return static_cast<const ResultBox*>(r)->r.facets[f].second.size(); }
const char* vo_result_facet_value(const void* r, size_t f, size_t i) { return static_cast<const ResultBox*>(r)->r.facets[f].second[i].first.c_str(); }
uint64_t vo_result_facet_count(const void* r, size_t f, size_t i) { return static_cast<const ResultBox*>(r)->r.facets[f].second[i].second; }
void vo_result_free(void* r) { delete static_cast<ResultBox*>(r); }

// CPU baseline: run the n requests (cycled `repeat` times) on `threads` host threads, one independent
// query per thread at a time (mirrors concurrent search() calls, server/rocket_server.rs:139-145).
// Returns wall seconds; per-query latencies (ns) are written to lat_ns[n*repeat] when non-null.
double vo_bench_search(const void* index, const char* const* jsons, const size_t* lens, size_t n, size_t repeat, int threads, uint64_t* lat_ns,
                       uint64_t* checksum) {
    // A query over 100 M documents materialises its operands' hit lists (tens of MB each, as the reference does).  glibc hands such
    // blocks back to the kernel on every free: with many threads the page faults of the next query then serialise on the process's
    // address-space lock (256 threads ran 4x SLOWER than 16).  Keep them in the threads' arenas instead.
    static const bool tuned = [] {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, -1);
        return true;
    }();
    (void)tuned;
    std::vector<Request> reqs;
    reqs.reserve(n);
    for (size_t i = 0; i < n; ++i) reqs.push_back(request_from_json_text(jsons[i], lens[i]));
    const Index& idx = *static_cast<const Index*>(index);
    size_t total = n * repeat;
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> sum{0};
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        uint64_t local = 0;
        while (true) {
            size_t i = next.fetch_add(1);
            if (i >= total) break;
            auto a = std::chrono::steady_clock::now();
            SearchResult r = search(reqs[i % n], idx);
            auto b = std::chrono::steady_clock::now();
            if (lat_ns) lat_ns[i] = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count());
            local += r.num_hits;
            for (auto& h : r.data) local += h.id;
        }
        sum += local;
    };
    std::vector<std::thread> ts;
    for (int t = 1; t < threads; ++t) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (checksum) *checksum = sum.load();
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- single-function entry points for the reference's known-answer vectors
static std::vector<SearchFieldResult> lists_from_flat(int nlists, const uint32_t* lens, const uint32_t* ids, const float* scores, const char* const* terms) {
    std::vector<SearchFieldResult> out(nlists);
    size_t off = 0;
    for (int l = 0; l < nlists; ++l) {
        for (uint32_t i = 0; i < lens[l]; ++i) {
            if (scores) out[l].hits_scores.push_back(Hit{ids[off + i], scores[off + i]});
            else out[l].hits_ids.push_back(ids[off + i]);
        }
        if (terms) out[l].request.terms = {terms[l]};
        off += lens[l];
    }
    return out;
}
static uint32_t hits_out(const SearchFieldResult& r, uint32_t* out_ids, float* out_scores) {
    if (out_scores) {
        for (size_t i = 0; i < r.hits_scores.size(); ++i) {
            out_ids[i] = r.hits_scores[i].id;
            out_scores[i] = r.hits_scores[i].score;
        }
        return uint32_t(r.hits_scores.size());
    }
    for (size_t i = 0; i < r.hits_ids.size(); ++i) out_ids[i] = r.hits_ids[i];
    return uint32_t(r.hits_ids.size());
}
uint32_t vo_op_intersect_hits_score(int n, const uint32_t* lens, const uint32_t* ids, const float* scores, uint32_t* out_ids, float* out_scores) {
    return hits_out(intersect_hits_score(lists_from_flat(n, lens, ids, scores, nullptr)), out_ids, out_scores);
}
uint32_t vo_op_union_hits_score(int n, const uint32_t* lens, const uint32_t* ids, const float* scores, const char* const* terms, uint32_t* out_ids,
                                float* out_scores) {
    return hits_out(union_hits_score(lists_from_flat(n, lens, ids, scores, terms)), out_ids, out_scores);
}
uint32_t vo_op_intersect_hits_ids(int n, const uint32_t* lens, const uint32_t* ids, uint32_t* out_ids) {
    return hits_out(intersect_hits_ids(lists_from_flat(n, lens, ids, nullptr, nullptr)), out_ids, nullptr);
}
uint32_t vo_op_union_hits_ids(int n, const uint32_t* lens, const uint32_t* ids, uint32_t* out_ids) {
    return hits_out(union_hits_ids(lists_from_flat(n, lens, ids, nullptr, nullptr)), out_ids, nullptr);
}
uint32_t vo_op_intersect_score_hits_with_ids(uint32_t nh, const uint32_t* ids, const float* scores, uint32_t nf, const uint32_t* filter, uint32_t* out_ids,
                                             float* out_scores) {
    uint32_t l1[1] = {nh}, l2[1] = {nf};
    auto a = lists_from_flat(1, l1, ids, scores, nullptr);
    auto b = lists_from_flat(1, l2, filter, nullptr, nullptr);
    return hits_out(intersect_score_hits_with_ids(std::move(a[0]), std::move(b[0])), out_ids, out_scores);
}
// boost lists: ids-only lists, boost value per list (NaN = default 2.0, boost.rs:393)
uint32_t vo_op_boost_hits_ids_vec_multi(uint32_t nh, const uint32_t* ids, const float* scores, int nlists, const uint32_t* lens, const uint32_t* bids,
                                        const float* boost_vals, uint32_t* out_ids, float* out_scores) {
    uint32_t l1[1] = {nh};
    auto a = lists_from_flat(1, l1, ids, scores, nullptr);
    auto b = lists_from_flat(nlists, lens, bids, nullptr, nullptr);
    for (int i = 0; i < nlists; ++i)
        if (boost_vals && boost_vals[i] == boost_vals[i]) b[i].request.boost = boost_vals[i];
    return hits_out(boost_hits_ids_vec_multi(std::move(a[0]), b), out_ids, out_scores);
}
// boost_fun: -1 none, 0 Log2, 1 Log10, 2 Multiply, 3 Add, 4 Replace
uint32_t vo_op_apply_boost_values_anchor(uint32_t nh, const uint32_t* ids, const float* scores, uint32_t nb, const uint32_t* bids, const float* bvals,
                                         int boost_fun, float param, int has_param, const char* expression, uint32_t* out_ids, float* out_scores) {
    uint32_t l1[1] = {nh};
    auto a = lists_from_flat(1, l1, ids, scores, nullptr);
    std::vector<Hit> b;
    for (uint32_t i = 0; i < nb; ++i) b.push_back(Hit{bids[i], bvals[i]});
    RequestBoostPart bp;
    if (boost_fun >= 0) bp.boost_fun = BoostFunction(boost_fun);
    if (has_param) bp.param = param;
    if (expression) bp.expression = std::string(expression);
    apply_boost_values_anchor(a[0], bp, b);
    return hits_out(a[0], out_ids, out_scores);
}
uint32_t vo_op_top_n_sort(uint32_t nh, const uint32_t* ids, const float* scores, uint32_t top_n, uint32_t* out_ids, float* out_scores) {
    std::vector<Hit> v;
    for (uint32_t i = 0; i < nh; ++i) v.push_back(Hit{ids[i], scores[i]});
    auto r = top_n_sort(std::move(v), top_n);
    for (size_t i = 0; i < r.size(); ++i) {
        out_ids[i] = r[i].id;
        out_scores[i] = r[i].score;
    }
    return uint32_t(r.size());
}
uint32_t vo_op_distance(const char* a, const char* b) { return distance(a, b); }
uint32_t vo_op_levenshtein(const char* a, const char* b, int transposition, int ci) {
    return levenshtein_cps(vqtext::decode_utf8(std::string(a)), vqtext::decode_utf8(std::string(b)), transposition != 0, ci != 0);
}
int vo_op_score_expression(const char* expr, float rank, float* out) {
    try {
        *out = score_expression(expr, rank);
        return 0;
    } catch (const VelociError& e) {
        return fail(e);
    }
}
float vo_op_default_score_for_distance(uint32_t d, int prefix) { return get_default_score_for_distance(uint8_t(d), prefix != 0); }
uint32_t vo_op_calculate_token_score(uint32_t pos, uint32_t occ, uint32_t ntok, int exact) { return calculate_token_score_for_entry(pos, occ, ntok, exact != 0); }
uint16_t vo_op_f32_to_f16(float f) { return f32_to_f16_bits(f); }
float vo_op_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }
uint32_t vo_op_steps_to_anchor(const char* path, char* out, size_t cap) {
    auto v = get_steps_to_anchor(path);
    std::string joined;
    for (auto& s : v) {
        joined += s;
        joined += '\n';
    }
    if (joined.size() + 1 <= cap) std::memcpy(out, joined.c_str(), joined.size() + 1);
    return uint32_t(v.size());
}

}  // extern "C"
