// search::Request wire structs (reference src/search/request/*.rs) and their serde-JSON parser.
// Wire-format plumbing shared by the product and the test oracle; no query arithmetic lives here.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "json.hpp"

namespace vqreq {

// error codes == include/veloci_amd.h
enum { OK = 0, ERR_INVALID_REQUEST = 1, ERR_FST_NOT_FOUND = 2, ERR_INDEX_NOT_FOUND = 3, ERR_UNSUPPORTED = 4, ERR_DEVICE = 5, ERR_INVALID_ARGUMENT = 6, ERR_JSON = 7 };

struct VelociError : std::runtime_error {  // src/error.rs:5-43
    int code;
    VelociError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

enum class BoostFunction { Log2, Log10, Multiply, Add, Replace };  // src/search/request/boost_request.rs:22-33

struct RequestBoostPart {  // src/search/request/boost_request.rs:3-20
    std::string path;
    std::optional<BoostFunction> boost_fun;
    std::optional<float> param;
    std::optional<std::vector<float>> skip_when_score;
    std::optional<std::string> expression;
    inline std::string key() const;
};

struct SearchRequestOptions {  // src/search/request/search_request.rs:103-119
    bool explain = false;
    std::optional<size_t> top, skip;
    std::optional<std::vector<RequestBoostPart>> boost;
    inline std::string key() const;
};

struct SnippetInfo {  // src/search/request/snippet_info.rs:1-39 (the defaults are DEFAULT_SNIPPETINFO)
    int64_t num_words_around_snippet = 5;
    std::string snippet_start_tag = "<b>", snippet_end_tag = "</b>", snippet_connector = " ... ";
    uint32_t max_snippets = 0xFFFFFFFFu;
};

struct RequestSearchPart {  // src/search/request/search_request.rs:127-179
    std::string path;
    std::vector<std::string> terms;
    std::optional<uint32_t> levenshtein_distance;
    bool starts_with = false;
    bool is_regex = false;
    std::optional<RequestBoostPart> token_value;
    std::optional<float> boost;
    std::optional<bool> ignore_case;
    std::optional<bool> snippet;
    bool has_snippet_info = false;
    SnippetInfo snippet_info;  // meaningful when has_snippet_info
    std::optional<size_t> top, skip;
    std::optional<SearchRequestOptions> options;
    inline std::string key() const;  // stands in for derive(PartialEq, Hash): equal keys <=> equal requests
};

struct SearchRequest;
struct SearchTree {  // src/search/request/search_request.rs:15-23
    std::vector<SearchRequest> queries;
    std::optional<SearchRequestOptions> options;
};
struct SearchRequest {  // src/search/request/search_request.rs:6-13
    enum Kind { Or, And, Search } kind = Search;
    SearchTree tree;
    RequestSearchPart part;
    const std::optional<SearchRequestOptions>& get_options() const { return kind == Search ? part.options : tree.options; }
};

struct FacetRequest {  // src/search/request/facet_request.rs:2-11
    std::string field;
    std::optional<size_t> top = 10;
};
struct RequestPhraseBoost {  // src/search/request/mod.rs:89-93
    RequestSearchPart search1, search2;
};

struct Request {  // src/search/request/mod.rs:15-87
    std::optional<SearchRequest> search_req;
    bool has_suggest = false;
    std::optional<std::vector<RequestSearchPart>> suggest;  // request/mod.rs:21-24: answered by suggest_multi (search_field.rs:194-219), not by search()
    std::optional<std::vector<RequestBoostPart>> boost;
    std::optional<std::vector<RequestSearchPart>> boost_term;
    std::optional<std::vector<FacetRequest>> facets;
    std::optional<std::vector<RequestPhraseBoost>> phrase_boosts;
    bool has_select = false;
    std::optional<SearchRequest> filter;
    std::optional<size_t> top = 10;  // default_top, src/search.rs:46-48
    std::optional<size_t> skip;
    bool why_found = false;
    bool text_locality = false;
    bool explain = false;
    // internal, never parsed: only hits ranking BELOW this key are returned (~0: no bound) — page p of a deep request (top + skip beyond
    // what one scan ranks) asks for what lies below the last key of page p-1
    uint64_t key_upper = ~0ull;
    bool exact_routes_only = false;  // (internal) second run of a request whose speculative route could not be confirmed: no k_scan_probe_or
};


// =====================================================================================
// Request keys (stand in for derive(PartialEq, Eq, Hash) on the request structs)
// =====================================================================================
inline void key_f(std::string& s, float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    char buf[16];
    std::snprintf(buf, sizeof buf, "%08x", b);
    s += buf;
}
inline void key_s(std::string& s, const std::string& v) {
    s += std::to_string(v.size());
    s += ':';
    s += v;
}
inline std::string RequestBoostPart::key() const {
    std::string s = "B{";
    key_s(s, path);
    s += boost_fun ? char('0' + int(*boost_fun)) : '-';
    if (param) key_f(s, *param);
    s += '|';
    if (skip_when_score) {
        s += '[';
        for (float f : *skip_when_score) key_f(s, f), s += ',';
        s += ']';
    }
    s += '|';
    if (expression) key_s(s, *expression);
    s += '}';
    return s;
}
inline std::string SearchRequestOptions::key() const {
    std::string s = "O{";
    s += explain ? '1' : '0';
    s += top ? std::to_string(*top) : "-";
    s += ',';
    s += skip ? std::to_string(*skip) : "-";
    s += ',';
    if (boost) {
        s += '[';
        for (auto& b : *boost) s += b.key();
        s += ']';
    }
    s += '}';
    return s;
}
inline std::string RequestSearchPart::key() const {
    std::string s = "P{";
    key_s(s, path);
    for (auto& t : terms) key_s(s, t);
    s += '|';
    s += levenshtein_distance ? std::to_string(*levenshtein_distance) : "-";
    s += starts_with ? 'S' : 's';
    s += is_regex ? 'R' : 'r';
    if (token_value) s += token_value->key();
    s += '|';
    if (boost) key_f(s, *boost);
    s += '|';
    s += ignore_case ? (*ignore_case ? '1' : '0') : '-';
    s += snippet ? (*snippet ? '1' : '0') : '-';
    s += has_snippet_info ? 'I' : 'i';
    if (has_snippet_info) {
        s += std::to_string(snippet_info.num_words_around_snippet) + "," + std::to_string(snippet_info.max_snippets);
        key_s(s, snippet_info.snippet_start_tag);
        key_s(s, snippet_info.snippet_end_tag);
        key_s(s, snippet_info.snippet_connector);
    }
    s += top ? std::to_string(*top) : "-";
    s += ',';
    s += skip ? std::to_string(*skip) : "-";
    s += ',';
    if (options) s += options->key();
    s += '}';
    return s;
}

// =====================================================================================
// serde-JSON -> Request (src/search/request/*.rs)
// =====================================================================================
[[noreturn]] inline void json_fail(const std::string& m) { throw VelociError(ERR_JSON, "JsonError: " + m); }

inline std::string j_string(const vqjson::Value& v, const char* what) {
    if (!v.is_string()) json_fail(std::string("invalid type for ") + what + ", expected a string");
    return v.str;
}
inline bool j_bool(const vqjson::Value& v, const char* what) {
    if (!v.is_bool()) json_fail(std::string("invalid type for ") + what + ", expected a boolean");
    return v.b;
}
inline float j_f32(const vqjson::Value& v, const char* what) {
    if (!v.is_number()) json_fail(std::string("invalid type for ") + what + ", expected a number");
    return float(v.num);  // serde_json: f64 parse, then `as f32`
}
inline size_t j_usize(const vqjson::Value& v, const char* what) {
    if (!v.is_number() || !v.is_integer || v.num < 0) json_fail(std::string("invalid type for ") + what + ", expected an unsigned integer");
    return size_t(v.num);
}
template <class T, class F>
inline std::optional<T> j_opt(const vqjson::Value& obj, const char* key, F f) {
    const vqjson::Value* v = obj.get(key);
    if (!v || v->is_null()) return std::nullopt;
    return f(*v, key);
}

inline RequestBoostPart boost_part_from_json(const vqjson::Value& v) {
    if (!v.is_object()) json_fail("RequestBoostPart: expected an object");
    RequestBoostPart b;
    const vqjson::Value* p = v.get("path");
    if (!p) json_fail("missing field `path`");
    b.path = j_string(*p, "path");
    if (const vqjson::Value* f = v.get("boost_fun"); f && !f->is_null()) {
        std::string n = j_string(*f, "boost_fun");
        if (n == "Log2") b.boost_fun = BoostFunction::Log2;
        else if (n == "Log10") b.boost_fun = BoostFunction::Log10;
        else if (n == "Multiply") b.boost_fun = BoostFunction::Multiply;
        else if (n == "Add") b.boost_fun = BoostFunction::Add;
        else if (n == "Replace") b.boost_fun = BoostFunction::Replace;
        else json_fail("unknown variant `" + n + "`, expected one of `Log2`, `Log10`, `Multiply`, `Add`, `Replace`");
    }
    b.param = j_opt<float>(v, "param", j_f32);
    if (const vqjson::Value* s = v.get("skip_when_score"); s && !s->is_null()) {
        if (!s->is_array()) json_fail("skip_when_score: expected a sequence");
        std::vector<float> out;
        for (auto& e : s->arr) out.push_back(j_f32(e, "skip_when_score"));
        b.skip_when_score = out;
    }
    b.expression = j_opt<std::string>(v, "expression", j_string);
    return b;
}

inline SearchRequestOptions options_from_json(const vqjson::Value& v) {
    if (!v.is_object()) json_fail("SearchRequestOptions: expected an object");
    SearchRequestOptions o;
    if (const vqjson::Value* e = v.get("explain"); e && !e->is_null()) o.explain = j_bool(*e, "explain");
    o.top = j_opt<size_t>(v, "top", j_usize);
    o.skip = j_opt<size_t>(v, "skip", j_usize);
    if (const vqjson::Value* b = v.get("boost"); b && !b->is_null()) {
        if (!b->is_array()) json_fail("boost: expected a sequence");
        std::vector<RequestBoostPart> out;
        for (auto& e : b->arr) out.push_back(boost_part_from_json(e));
        o.boost = out;
    }
    return o;
}

inline SnippetInfo snippet_info_from_json(const vqjson::Value& sv) {  // src/search/request/snippet_info.rs:1-13: every field has a serde default
    if (!sv.is_object()) json_fail("SnippetInfo: expected an object");
    const vqjson::Value* s = &sv;
    SnippetInfo si;
    if (const vqjson::Value* n = s->get("num_words_around_snippet")) {
        if (!n->is_number() || !n->is_integer || n->num < -9223372036854775808.0 || n->num >= 9223372036854775808.0)
            json_fail("invalid type for num_words_around_snippet, expected i64");
        si.num_words_around_snippet = int64_t(n->num);
    }
    if (const vqjson::Value* t = s->get("snippet_start_tag")) si.snippet_start_tag = j_string(*t, "snippet_start_tag");
    if (const vqjson::Value* t = s->get("snippet_end_tag")) si.snippet_end_tag = j_string(*t, "snippet_end_tag");
    if (const vqjson::Value* t = s->get("snippet_connector")) si.snippet_connector = j_string(*t, "snippet_connector");
    if (const vqjson::Value* m = s->get("max_snippets")) {
        const size_t d = j_usize(*m, "max_snippets");
        if (d > 0xFFFFFFFFull) json_fail("invalid value for max_snippets, expected u32");
        si.max_snippets = uint32_t(d);
    }
    return si;
}

inline RequestSearchPart search_part_from_json(const vqjson::Value& v) {
    if (!v.is_object()) json_fail("RequestSearchPart: expected an object");
    RequestSearchPart p;
    const vqjson::Value* path = v.get("path");
    if (!path) json_fail("missing field `path`");
    p.path = j_string(*path, "path");
    const vqjson::Value* terms = v.get("terms");
    if (!terms) json_fail("missing field `terms`");
    if (!terms->is_array()) json_fail("terms: expected a sequence");
    for (auto& t : terms->arr) p.terms.push_back(j_string(t, "terms"));
    if (const vqjson::Value* l = v.get("levenshtein_distance"); l && !l->is_null()) {
        const size_t d = j_usize(*l, "levenshtein_distance");
        if (d > 0xFFFFFFFFull) json_fail("invalid value for levenshtein_distance, expected u32");
        p.levenshtein_distance = uint32_t(d);
    }
    if (const vqjson::Value* s = v.get("starts_with"); s) p.starts_with = j_bool(*s, "starts_with");
    if (const vqjson::Value* s = v.get("is_regex"); s) p.is_regex = j_bool(*s, "is_regex");
    if (const vqjson::Value* t = v.get("token_value"); t && !t->is_null()) p.token_value = boost_part_from_json(*t);
    p.boost = j_opt<float>(v, "boost", j_f32);
    p.ignore_case = j_opt<bool>(v, "ignore_case", j_bool);
    p.snippet = j_opt<bool>(v, "snippet", j_bool);
    if (const vqjson::Value* s = v.get("snippet_info"); s && !s->is_null()) {
        p.has_snippet_info = true;
        p.snippet_info = snippet_info_from_json(*s);
    }
    p.top = j_opt<size_t>(v, "top", j_usize);
    p.skip = j_opt<size_t>(v, "skip", j_usize);
    if (const vqjson::Value* o = v.get("options"); o && !o->is_null()) p.options = options_from_json(*o);
    return p;
}

inline SearchRequest search_request_from_json(const vqjson::Value& v) {
    if (!v.is_object() || v.obj.size() != 1) json_fail("SearchRequest: expected a map with a single key (`or`, `and`, `search`)");
    const std::string& tag = v.obj[0].first;
    const vqjson::Value& body = v.obj[0].second;
    SearchRequest r;
    if (tag == "search") {
        r.kind = SearchRequest::Search;
        r.part = search_part_from_json(body);
        return r;
    }
    if (tag == "or") r.kind = SearchRequest::Or;
    else if (tag == "and") r.kind = SearchRequest::And;
    else json_fail("unknown variant `" + tag + "`, expected one of `or`, `and`, `search`");
    if (!body.is_object()) json_fail("SearchTree: expected an object");
    const vqjson::Value* q = body.get("queries");
    if (!q) json_fail("missing field `queries`");
    if (!q->is_array()) json_fail("queries: expected a sequence");
    for (auto& e : q->arr) r.tree.queries.push_back(search_request_from_json(e));
    if (const vqjson::Value* o = body.get("options"); o && !o->is_null()) r.tree.options = options_from_json(*o);
    return r;
}

inline Request request_from_json(const vqjson::Value& v) {
    if (!v.is_object()) json_fail("Request: expected an object");
    Request r;
    if (const vqjson::Value* s = v.get("search_req"); s && !s->is_null()) r.search_req = search_request_from_json(*s);
    if (const vqjson::Value* s = v.get("suggest"); s && !s->is_null()) {
        if (!s->is_array()) json_fail("suggest: expected a sequence");
        std::vector<RequestSearchPart> out;
        for (auto& e : s->arr) out.push_back(search_part_from_json(e));
        r.suggest = out;
        r.has_suggest = true;
    }
    if (const vqjson::Value* b = v.get("boost"); b && !b->is_null()) {
        if (!b->is_array()) json_fail("boost: expected a sequence");
        std::vector<RequestBoostPart> out;
        for (auto& e : b->arr) out.push_back(boost_part_from_json(e));
        r.boost = out;
    }
    if (const vqjson::Value* b = v.get("boost_term"); b && !b->is_null()) {
        if (!b->is_array()) json_fail("boost_term: expected a sequence");
        std::vector<RequestSearchPart> out;
        for (auto& e : b->arr) out.push_back(search_part_from_json(e));
        r.boost_term = out;
    }
    if (const vqjson::Value* f = v.get("facets"); f && !f->is_null()) {
        if (!f->is_array()) json_fail("facets: expected a sequence");
        std::vector<FacetRequest> out;
        for (auto& e : f->arr) {
            if (!e.is_object()) json_fail("FacetRequest: expected an object");
            FacetRequest fr;
            const vqjson::Value* fld = e.get("field");
            if (!fld) json_fail("missing field `field`");
            fr.field = j_string(*fld, "field");
            if (const vqjson::Value* t = e.get("top"); t) {  // present: null -> None, number -> Some
                if (t->is_null()) fr.top = std::nullopt;
                else fr.top = j_usize(*t, "top");
            }
            out.push_back(fr);
        }
        r.facets = out;
    }
    if (const vqjson::Value* p = v.get("phrase_boosts"); p && !p->is_null()) {
        if (!p->is_array()) json_fail("phrase_boosts: expected a sequence");
        std::vector<RequestPhraseBoost> out;
        for (auto& e : p->arr) {
            if (!e.is_object()) json_fail("RequestPhraseBoost: expected an object");
            const vqjson::Value* s1 = e.get("search1");
            const vqjson::Value* s2 = e.get("search2");
            if (!s1) json_fail("missing field `search1`");
            if (!s2) json_fail("missing field `search2`");
            out.push_back({search_part_from_json(*s1), search_part_from_json(*s2)});
        }
        r.phrase_boosts = out;
    }
    if (const vqjson::Value* s = v.get("select"); s && !s->is_null()) {  // Option<Vec<String>>
        if (!s->is_array()) json_fail("select: expected a sequence");
        for (auto& e : s->arr) (void)j_string(e, "select");
        r.has_select = true;
    }
    if (const vqjson::Value* f = v.get("filter"); f && !f->is_null()) r.filter = search_request_from_json(*f);
    if (const vqjson::Value* t = v.get("top"); t) {
        if (t->is_null()) r.top = std::nullopt;
        else r.top = j_usize(*t, "top");
    }
    if (const vqjson::Value* s = v.get("skip"); s && !s->is_null()) r.skip = j_usize(*s, "skip");
    if (const vqjson::Value* w = v.get("why_found"); w) r.why_found = j_bool(*w, "why_found");
    if (const vqjson::Value* w = v.get("text_locality"); w) r.text_locality = j_bool(*w, "text_locality");
    if (const vqjson::Value* w = v.get("explain"); w) r.explain = j_bool(*w, "explain");
    return r;
}


inline Request request_from_json_text(const char* s, size_t n) {
    try {
        return request_from_json(vqjson::parse(s, n));
    } catch (const vqjson::ParseError& e) {
        throw VelociError(ERR_JSON, std::string("JsonError: ") + e.what());
    }
}

}  // namespace vqreq
