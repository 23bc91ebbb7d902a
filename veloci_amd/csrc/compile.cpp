// Query compiler: search::Request -> device program (lists + postfix ops + sink stages).
//
// Mirrors what the reference does while it creates and wires its plan
// (src/plan_creator/execution_plan.rs:91-534, plan_steps.rs:137-345) and the host-only bookkeeping of
// search() (src/search.rs:143-228): leaf de-duplication, filter / boost / phrase wiring, term-id
// bookkeeping for text locality, path algebra.  Everything that touches postings, scores or hit
// lists is NOT done here — it is encoded for k_tile_scan.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <regex>
#include <set>

#include "engine.hpp"
#include "text.hpp"

namespace vq {
std::atomic<uint64_t> g_compile_ns[16];  // 0 lookup_terms, 1 resolve_boost_1n, 2 emit_boost_1n rest, 3 leaf lists, 4 the longest single compile_query, 5 total;
                                         // inside 1: 6 value-id gather, 7 sort, 8 (anchor, value) pairs; 9 order check + layers of a resolved list

using namespace vqreq;

namespace {

bool ends_with(const std::string& s, const char* suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && std::memcmp(s.data() + s.size() - n, suf, n) == 0;
}

[[noreturn]] void unsupported(const std::string& what) {
    throw VelociError(ERR_UNSUPPORTED, "unsupported on the MI355X query path: " + what);
}

// reference src/util.rs:147-162
std::vector<std::string> get_steps_to_anchor(const std::string& path) {
    std::vector<std::string> paths;
    std::string current;
    size_t start = 0;
    while (true) {
        size_t dot = path.find('.', start);
        std::string part = path.substr(start, dot == std::string::npos ? std::string::npos : dot - start);
        if (!current.empty()) current += ".";
        current += part;
        if (ends_with(part, "[]")) paths.push_back(current);
        if (dot == std::string::npos) break;
        start = dot + 1;
    }
    paths.push_back(path + TEXTINDEX);
    return paths;
}

// search_field.rs:27-33
float default_score_for_distance(uint8_t distance, bool prefix_matches) {
    if (prefix_matches) return 2.0f / (std::log2(float(distance) + 1.0f) + 0.2f);
    return 2.0f / (float(distance) + 0.2f);
}

bool cp_eq(uint32_t a, uint32_t b, bool ci) { return a == b || (ci && vqtext::lower_cp(a) == vqtext::lower_cp(b)); }

// Distance the reference scores a dictionary hit with (search_field.rs:691-732): the scoring automaton is built
// over the lower-cased term with transpositions at cost one; beyond its maximum it falls back to the plain
// character Levenshtein distance in u8 (255 for strings of 255 bytes or more).
uint8_t scoring_distance(const std::string& lower_hit, const std::string& lower_term, uint32_t dfa_max) {
    if (lower_hit == lower_term) return 0;  // (both distances of equal strings: the exact-match leaf, the common case)
    const auto h = vqtext::decode_utf8(lower_hit), t = vqtext::decode_utf8(lower_term);
    const size_t n = h.size(), m = t.size(), w = m + 1;
    std::vector<uint32_t> osa((n + 1) * w), lev((n + 1) * w);
    for (size_t j = 0; j <= m; ++j) osa[j] = lev[j] = uint32_t(j);
    for (size_t i = 1; i <= n; ++i) {
        osa[i * w] = lev[i * w] = uint32_t(i);
        for (size_t j = 1; j <= m; ++j) {
            const uint32_t sub = h[i - 1] == t[j - 1] ? 0u : 1u;
            lev[i * w + j] = std::min({lev[(i - 1) * w + j] + 1, lev[i * w + j - 1] + 1, lev[(i - 1) * w + j - 1] + sub});
            uint32_t v = std::min({osa[(i - 1) * w + j] + 1, osa[i * w + j - 1] + 1, osa[(i - 1) * w + j - 1] + sub});
            if (i > 1 && j > 1 && h[i - 1] == t[j - 2] && h[i - 2] == t[j - 1]) v = std::min(v, osa[(i - 2) * w + j - 2] + 1);
            osa[i * w + j] = v;
        }
    }
    if (osa[n * w + m] <= dfa_max) return uint8_t(osa[n * w + m]);
    if (lower_hit.size() >= 255 || lower_term.size() >= 255) return 255;
    return uint8_t(lev[n * w + m]);
}

// levenshtein_distance after the clamp of search_field.rs:285-287 (0 when absent; the empty-term error is raised by the compiler)
uint32_t clamped_lev(const RequestSearchPart& p) {
    if (!p.levenshtein_distance) return 0;
    const size_t chars = vqtext::decode_utf8(vqtext::to_lower_utf8(p.terms.empty() ? std::string() : p.terms[0])).size();
    if (chars == 0) return 0;
    return std::min<uint32_t>(*p.levenshtein_distance, uint32_t(chars) - 1);
}

// expression.rs:48-95 — "x op y", x/y = $SCORE or a float
void parse_expression(const std::string& expression, DColBoost& cb) {
    struct Tok {
        int kind;  // 0 score, 1 float, 2.. operators (2 div, 3 mul, 4 add, 5 sub)
        float val;
    };
    std::vector<Tok> ops;
    std::string current;
    auto try_float = [](const std::string& s, float& out) {
        if (s.empty()) return false;
        char* end = nullptr;
        out = std::strtof(s.c_str(), &end);
        return end && *end == 0 && end != s.c_str();
    };
    for (char ch : expression) {
        if (ch == ' ') {
            float v;
            if (try_float(current, v)) ops.push_back({1, v});
            current.clear();
        } else current.push_back(ch);
        if (current == "+") ops.push_back({4, 0}), current.clear();
        else if (current == "-") ops.push_back({5, 0}), current.clear();
        else if (current == "/") ops.push_back({2, 0}), current.clear();
        else if (current == "*") ops.push_back({3, 0}), current.clear();
        else if (current == "$SCORE") ops.push_back({0, 0}), current.clear();
    }
    float v;
    if (try_float(current, v)) ops.push_back({1, v});
    if (ops.size() < 3 || ops[0].kind > 1 || ops[2].kind > 1 || ops[1].kind < 2)
        throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"bad score expression " + expression + "\" ");
    cb.expr_lkind = ops[0].kind;
    cb.expr_lval = ops[0].val;
    cb.expr_rkind = ops[2].kind;
    cb.expr_rval = ops[2].val;
    cb.expr_op = ops[1].kind == 2 ? EX_DIV : ops[1].kind == 3 ? EX_MUL : ops[1].kind == 4 ? EX_ADD : EX_SUB;
}

struct Leaf {  // PlanStepFieldSearchToTokenIds + its result (execution_plan.rs:16-44, plan_steps.rs:137-148)
    const RequestSearchPart* part = nullptr;  // the request's own part (it outlives the compilation)
    std::string key;                          // part->key(), built once
    const PostingStore* store = nullptr;      // "<path>.to_anchor_id_score", looked up once
    std::string path;  // with ".textindex"
    bool get_scores = false, get_ids = false, store_term_id_hits = false;
    bool return_term = false, return_term_lowercase = false, store_term_texts = false;  // execution_plan.rs:16-44
    std::vector<std::pair<uint32_t, std::string>> terms;  // term id -> text (search_field.rs:347-353), ascending ids
    bool computed = false;
    std::vector<std::pair<uint32_t, float>> hits_scores;  // (term id, term score), ascending term id
    std::vector<uint32_t> hits_ids;                        // term ids
    std::map<uint32_t, ExplainRecs> explain;               // options.explain: the dictionary result's records per term id (search_field.rs:334-343)
};
// Ascending, duplicate-free — for id lists that can reach 10^5 entries per leaf (the value ids behind a prefix leaf's 1:n boost): a bitmap over
// the ids' span when it is dense enough (one pass to mark, one to read back), LSD radix passes otherwise; short lists take std::sort.
void sort_unique_u32(std::vector<uint32_t>& v) {
    const size_t n = v.size();
    if (n < 2048) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        return;
    }
    uint32_t lo = v[0], hi = v[0];
    for (uint32_t x : v) {
        lo = std::min(lo, x);
        hi = std::max(hi, x);
    }
    const uint64_t span = uint64_t(hi) - lo + 1, words = (span + 63) / 64;
    constexpr size_t kKeepBytes = 8u << 20;  // scratch a compile thread keeps between calls; larger buffers are handed back
    if (words <= n) {                        // (the bitmap is at most twice the ids' own size)
        thread_local std::vector<uint64_t> bits;
        struct Trim {
            std::vector<uint64_t>& v;
            ~Trim() {
                if (v.capacity() * sizeof(uint64_t) > kKeepBytes) std::vector<uint64_t>().swap(v);
            }
        } trim{bits};
        bits.assign(words, 0);
        for (uint32_t x : v) bits[(x - lo) >> 6] |= 1ull << ((x - lo) & 63);
        size_t k = 0;
        for (uint64_t w = 0; w < words; ++w) {
            uint64_t b = bits[w];
            while (b) {
                v[k++] = lo + uint32_t(w * 64) + uint32_t(__builtin_ctzll(b));
                b &= b - 1;
            }
        }
        v.resize(k);
        return;
    }
    thread_local std::vector<uint32_t> tmp;
    struct TrimTmp {
        std::vector<uint32_t>& v;
        ~TrimTmp() {
            if (v.capacity() * sizeof(uint32_t) > kKeepBytes) std::vector<uint32_t>().swap(v);
        }
    } trim_tmp{tmp};
    tmp.resize(n);
    uint32_t* a = v.data();
    uint32_t* b = tmp.data();
    for (uint32_t shift = 0; shift < 32 && (span - 1) >> shift; shift += 11) {
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; ++i) ++count[(((a[i] - lo) >> shift) & 2047u) + 1];
        for (int i = 0; i < 2048; ++i) count[i + 1] += count[i];
        for (size_t i = 0; i < n; ++i) b[count[((a[i] - lo) >> shift) & 2047u]++] = a[i];
        std::swap(a, b);
    }
    if (a != v.data()) std::memcpy(v.data(), a, n * sizeof(uint32_t));
    v.erase(std::unique(v.begin(), v.end()), v.end());
}

// VQ_TIMING: where request compilation spends its time (summed over threads, printed per batch by exec.cpp)
struct PhaseTimer {
    static bool on() {
        static const bool v = std::getenv("VQ_TIMING") != nullptr;
        return v;
    }
    int slot;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(int s) : slot(s) {
        if (on()) t0 = std::chrono::steady_clock::now();
    }
    ~PhaseTimer() {
        if (!on()) return;
        const uint64_t ns = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
        g_compile_ns[slot] += ns;
        if (slot == 5) {  // the longest single request: what bounds a parallel pass from below
            uint64_t cur = g_compile_ns[4].load();
            while (ns > cur && !g_compile_ns[4].compare_exchange_weak(cur, ns)) {
            }
        }
    }
};
static bool part_explains(const RequestSearchPart& p) { return p.options && p.options->explain; }  // search_request.rs:182-184

struct NodeInfo {
    std::string label;           // request.terms[0] carried by the node's result (set_op.rs:122,215,439)
    bool label_known = true;
    bool len_known = false;      // result length known without executing (needed for set_op.rs:388-393)
    uint64_t glen = 0;           // that length (unsharded)
    std::vector<uint32_t> cover; // lists that cover the node's result docs
    uint64_t cover_len = 0;      // shard-local entries of the cover
    bool emitted = false;        // false: node produced no op of its own (single child / passthrough)
    uint32_t node_id = UINT32_MAX;  // preorder number inside search_req (stable between compilation passes)
    int root_op = -1;               // index of the op whose presence is this node's result
    std::vector<int> ex_nodes;      // explain: the node(s) of the ExplainPlan behind this result (several: the members of a fused leaf) ...
    std::vector<size_t> ex_pos;     // ... and their positions among the parent's operands
};
struct CountReq {
    uint32_t node_id;
    int root_op;
};

struct Compiler {
    const Index& idx;
    const Request& req;
    CompiledQuery cq;
    std::map<std::string, Leaf> cache;  // FieldRequestCache
    std::vector<std::pair<const RequestSearchPart*, Leaf*>> by_address;
    std::map<std::string, std::map<std::string, std::vector<uint32_t>>> term_id_hits;  // path -> term -> term ids
    uint32_t max_depth = 0;

    const FuzzyTable* fuzzy = nullptr;
    const UnionTable* unions = nullptr;
    const RangeTable* ranges = nullptr;
    const LocalityTable* localities = nullptr;
    Boost1nCache* boost_cache = nullptr;
    const QueryCounts* counts = nullptr;
    uint32_t next_node = 0;
    std::vector<CountReq> count_reqs;   // operands whose result sizes a count pre-pass must measure
    std::vector<CountReq> maybe_reqs;   // ... only if some OR needs the label of a two-operand AND
    bool label_wanted = false;
    std::vector<const BoostColumn*> col_store;  // the boost column behind cq.cols[k] (request-level boosts): its value range bounds the boost's factor
    bool explain_on = false;                 // every leaf of the score tree carries options.explain (execution_plan.rs:46-85 after request.explain)
    std::shared_ptr<ExplainPlan> xplan;

    Compiler(const Index& i, const Request& r, const FuzzyTable* f) : idx(i), req(r), fuzzy(f) {}

    // ------------------------------------------------------------ explain plan
    int ex_leaf(const Leaf* l) {  // l == nullptr: the empty result of an and/or without operands
        ExplainNode n;
        n.kind = XP_LEAF;
        n.list_begin = uint32_t(xplan->lists.size());
        if (l) {
            const PostingStore& ps = posting_store(*l);
            for (auto& [tid, score] : l->hits_scores) {  // resolve_token_to_anchor walks the dictionary hits in this order (search_field.rs:419)
                ExList x{};
                if (tid < ps.num_tokens) {
                    x.docs = ps.docs.as<uint32_t>() + ps.start[tid];
                    x.scores = ps.scores.as<uint16_t>() + ps.start[tid];
                    x.len = ps.len[tid];
                }
                x.term_score = score;
                xplan->lists.push_back(x);
                n.list_term.push_back(tid);
            }
            n.term_records = l->explain;
        }
        n.list_count = uint32_t(xplan->lists.size()) - n.list_begin;
        xplan->nodes.push_back(std::move(n));
        return int(xplan->nodes.size()) - 1;
    }
    void ex_emit(int node, uint32_t& depth, uint32_t& max_depth_seen) {
        ExplainNode& n = xplan->nodes[size_t(node)];
        ExOp op{};
        op.kind = n.kind;
        if (n.kind == XP_LEAF) {
            op.a = n.list_begin;
            op.b = n.list_count;
        } else {
            for (int c : n.children) ex_emit(c, depth, max_depth_seen);
            op.nchild = uint32_t(n.children.size());
            op.a = uint32_t(xplan->aux.size());
            xplan->aux.insert(xplan->aux.end(), n.order.begin(), n.order.end());  // AND: summation order; OR: the operands' term slots
            if (n.kind == XP_OR) op.b = uint32_t(*std::max_element(n.order.begin(), n.order.end())) + 1u;
            depth -= op.nchild;
        }
        n.op = int(xplan->ops.size());
        xplan->ops.push_back(op);
        max_depth_seen = std::max(max_depth_seen, ++depth);
    }

    // ------------------------------------------------------------ leaves
    void add_to_cache(const RequestSearchPart& part, bool ids_only) {  // execution_plan.rs:108-130
        auto [it, fresh] = cache.try_emplace(part.key());
        Leaf& l = it->second;
        if (fresh) {
            l.part = &part;
            l.key = it->first;
        }
        l.get_ids |= ids_only;
        l.get_scores |= !ids_only;
        by_address.emplace_back(&part, &l);  // (std::map nodes do not move)
    }
    void collect(const SearchRequest& r, bool ids_only) {
        if (r.kind == SearchRequest::Search) add_to_cache(r.part, ids_only);
        else
            for (auto& q : r.tree.queries) collect(q, ids_only);
    }
    Leaf& leaf(const RequestSearchPart& part) {
        for (auto& pr : by_address)  // the parts of this request, by identity: building a key costs more than compiling a leaf
            if (pr.first == &part) return *pr.second;
        auto it = cache.find(part.key());
        if (it == cache.end()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"PlanCreator: Could not find request in field_search_cache\" ");
        return it->second;
    }
    void flag_tree(const SearchRequest& r) {  // execution_plan.rs:401-418
        if (r.kind == SearchRequest::Search) leaf(r.part).store_term_id_hits |= (req.why_found || req.text_locality);
        else
            for (auto& q : r.tree.queries) flag_tree(q);
    }
    void flag_tree_texts(const SearchRequest& r) {  // execution_plan.rs:416
        if (r.kind == SearchRequest::Search) leaf(r.part).store_term_texts |= req.why_found;
        else
            for (auto& q : r.tree.queries) flag_tree_texts(q);
    }

    // Regex leaf (search_field.rs:72-83), a HOST fallback for the dictionary side only (the postings of the matched terms stay on the device):
    // regex-automata 0.1.9's dense DFA walks the term's bytes, unanchored at the start (its builder acts as if the pattern began with
    // `(?s:.)*?`), and the term is accepted when the walk ENDS in a match state; `starts_with` accepts once any prefix did.  Restated over code
    // points with std::wregex (ECMAScript grammar: the common subset of the two syntaxes; case-insensitivity beyond ASCII follows the C locale).
    std::vector<uint32_t> regex_candidates(const Dictionary& dict, const RequestSearchPart& p) {
        auto widen = [](const std::string& u8) {
            std::wstring w;
            for (uint32_t cp : vqtext::decode_utf8(u8)) w.push_back(wchar_t(cp));
            return w;
        };
        std::wregex whole, anywhere;
        try {
            const auto flags = std::regex::ECMAScript | (p.ignore_case.value_or(true) ? std::regex::icase : std::regex::ECMAScript);
            const std::wstring pat = widen(p.terms[0]);
            whole = std::wregex(L"[\\s\\S]*?(?:" + pat + L")", flags);
            anywhere = std::wregex(pat, flags);
        } catch (const std::regex_error& e) {
            throw VelociError(ERR_INVALID_REQUEST, std::string("InvalidRequest: \"regex ") + e.what() + "\" ");
        }
        std::vector<uint32_t> out;
        for (uint32_t id = 0; id < dict.terms.size(); ++id) {
            const std::wstring w = widen(dict.terms[id]);
            if (p.starts_with ? std::regex_search(w, anywhere) : std::regex_match(w, whole)) out.push_back(id);
        }
        return out;
    }

    // get_term_ids_in_field (search_field.rs:277-398) — dictionary side only
    void lookup_terms(const Index& idx, Leaf& l, bool get_scores, bool get_ids) {
        PhaseTimer pt(0);
        const RequestSearchPart& p = *l.part;
        // (`snippet` / `snippet_info` are not looked at here: a search resolves token hits with resolve_token_hits_to_text_id_ids_only, plan_steps.rs:184 —
        //  snippets are made by search_field::highlight alone, search_field.rs:243 = vq_highlight_json)
        if (p.terms.empty()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"terms is empty\" ");
        l.path = p.path;
        if (!ends_with(l.path, TEXTINDEX)) l.path += TEXTINDEX;
        const std::string lower_term = vqtext::to_lower_utf8(p.terms[0]);
        uint32_t lev = 0;
        if (p.levenshtein_distance) {  // :285-287
            const auto lower_cps = vqtext::decode_utf8(lower_term);
            if (lower_cps.empty()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"empty term with levenshtein_distance\" ");
            lev = std::min<uint32_t>(*p.levenshtein_distance, uint32_t(lower_cps.size()) - 1);
        }
        auto dit = idx.dict.find(l.path);
        if (dit == idx.dict.end()) throw VelociError(ERR_FST_NOT_FOUND, "field does not exist " + l.path + " (fst not found)");
        const Dictionary& dict = dit->second;
        const bool ci = p.ignore_case.value_or(true);  // search_field.rs:88
        const bool query_ascii = vqtext::is_ascii(p.terms[0].data(), p.terms[0].size());
        std::vector<uint32_t> query_cps;  // decoded when a comparison needs code points
        std::vector<uint32_t> cand;
        const FuzzyProbe* probe = nullptr;
        const bool regex = p.is_regex;
        const bool scan = !regex && (lev != 0 || p.starts_with);
        if (regex) cand = regex_candidates(dict, p);
        else if (scan) {  // match set computed on the device before compilation (k_dict_scan), ascending == FST stream order
            const FuzzyProbe* fp = nullptr;
            if (fuzzy) {
                auto it = fuzzy->find(fuzzy_key(p));
                if (it != fuzzy->end()) fp = &it->second;
            }
            if (!fp) unsupported("levenshtein_distance > 0 / starts_with without a dictionary scan (internal)");
            if (fp->status != 0) throw VelociError(fp->status, fp->error);
            cand = fp->matches;
            probe = fp;
        } else if (ci) {
            auto it = dict.lower_map.find(lower_term);
            if (it != dict.lower_map.end()) cand = it->second;
        } else {
            auto it = std::lower_bound(dict.terms.begin(), dict.terms.end(), p.terms[0]);
            if (it != dict.terms.end() && *it == p.terms[0]) cand.push_back(uint32_t(it - dict.terms.begin()));
        }
        const bool limit_result = p.top.has_value();  // :292
        const size_t top_n_search = p.top.value_or(10) + p.skip.value_or(0);
        float worst_score = -std::numeric_limits<float>::max();
        const bool check_prefix = p.starts_with || lev != 0;  // :302
        for (size_t ci_ = 0; ci_ < cand.size(); ++ci_) {  // ascending ids == FST stream order
            const uint32_t id = cand[ci_];
            if (!scan && !regex) {
                const std::string& cand_term = dict.terms[id];
                bool eq = true;
                if (query_ascii && vqtext::is_ascii(cand_term.data(), cand_term.size())) {  // bytes are code points
                    eq = cand_term.size() == p.terms[0].size();
                    for (size_t i = 0; i < cand_term.size() && eq; ++i) eq = cp_eq(uint8_t(cand_term[i]), uint8_t(p.terms[0][i]), ci);
                } else {
                    const auto cps = vqtext::decode_utf8(cand_term);
                    if (query_cps.empty() && !p.terms[0].empty()) query_cps = vqtext::decode_utf8(p.terms[0]);
                    eq = cps.size() == query_cps.size();
                    for (size_t i = 0; i < cps.size() && eq; ++i) eq = cp_eq(cps[i], query_cps[i], ci);
                }
                if (!eq) continue;
            }
            if (get_ids) l.hits_ids.push_back(id);
            if (get_scores) {  // :304-354
                float score;
                if (probe) score = probe->scores[ci_];
                else {
                    const std::string lower_hit = vqtext::to_lower_utf8(dict.terms[id]);
                    const bool prefix_matches = check_prefix && lower_hit.compare(0, lower_term.size(), lower_term) == 0 && lower_hit.size() >= lower_term.size();
                    score = default_score_for_distance(scoring_distance(lower_hit, lower_term, lev), prefix_matches);
                }
                if (limit_result) {
                    if (score < worst_score) continue;  // (:324-327 returns before the term text is recorded)
                    if (!l.hits_scores.empty() && l.hits_scores.size() == top_n_search + 200) {  // sort.rs:24-34
                        std::sort(l.hits_scores.begin(), l.hits_scores.end(), [](auto& a, auto& b) { return a.second == b.second ? a.first > b.first : a.second > b.second; });
                        l.hits_scores.resize(top_n_search);
                        worst_score = l.hits_scores.back().second;
                    }
                }
                l.hits_scores.push_back({id, score});
                if (part_explains(p)) {  // :334-343 (kept for hits a later top-n cut drops again, as in the reference)
                    ExplainRec e;
                    e.kind = ExplainRec::LevenshteinScore;
                    e.a = score;
                    e.term_id = id;
                    e.text = dict.terms[id];
                    l.explain[id] = ExplainRecs{e};
                }
            }
            if (l.return_term || l.store_term_texts)  // :347-353
                l.terms.push_back({id, l.return_term_lowercase ? vqtext::to_lower_utf8(dict.terms[id]) : dict.terms[id]});
        }
        if (p.boost)  // :359-364
            for (auto& h : l.hits_scores) h.second *= *p.boost;
        if (limit_result) {  // :373-376 (the reference's sort is unstable: tie order is unspecified there)
            std::stable_sort(l.hits_scores.begin(), l.hits_scores.end(), [](auto& a, auto& b) { return a.second > b.second; });
            if (l.hits_scores.size() > top_n_search) l.hits_scores.resize(top_n_search);
        }
    }
    // token_value (search_field.rs:391-395): add_boost (boost.rs:470-504) over the matched TERMS — the boost column is keyed by term id
    // ("<path>.textindex.token_values.boost_valid_to_value", create/token_values_to_tokens.rs) and shapes the term scores before any
    // posting is read.  Host arithmetic (glibc log10f / log2f, as the reference links them).
    void apply_token_value(const RequestBoostPart& tv, Leaf& l) {
        const std::string path = tv.path + TEXTINDEX + TOKEN_VALUES + BOOST_VALID_TO_VALUE;
        auto bit = idx.boost.find(path);
        if (bit == idx.boost.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
        DColBoost cb{};
        fill_boost_params(cb, tv);
        for (auto& h : l.hits_scores) {
            bool skip = false;
            for (uint32_t s = 0; s < cb.nskip; ++s) skip = skip || std::fabs(cb.skip[s] - h.second) < 0.00001f;
            float v;
            if (skip || !bit->second.host_value(h.first, &v)) continue;
            const float vp = v + cb.param;
            float score = h.second;
            switch (cb.fun) {  // apply_boost, boost.rs:283-377
                case BF_LOG10: score *= std::log10(vp); break;
                case BF_LOG2: score *= std::log2(vp); break;
                case BF_MULTIPLY: score *= vp; break;
                case BF_ADD: score += vp; break;
                case BF_REPLACE: score = vp; break;
                default: break;
            }
            if (cb.expr_op != EX_NONE) {
                const float a = cb.expr_lkind == 0 ? v : cb.expr_lval, b = cb.expr_rkind == 0 ? v : cb.expr_rval;
                score += cb.expr_op == EX_DIV ? a / b : cb.expr_op == EX_MUL ? a * b : cb.expr_op == EX_ADD ? a + b : a - b;
            }
            if (part_explains(*l.part)) {  // boost.rs:297-300, 371-374
                ExplainRec e;
                e.kind = ExplainRec::Boost;
                if (cb.fun == BF_LOG10) {
                    e.a = std::log10(vp);
                    l.explain[h.first].push_back(e);
                }
                e.a = score;
                l.explain[h.first].push_back(e);
            }
            h.second = score;
        }
    }
    Leaf& field_result(const RequestSearchPart& part) {
        Leaf& l = leaf(part);
        if (!l.computed) {
            lookup_terms(idx, l, l.get_scores, l.get_ids);
            if (l.store_term_id_hits && !l.hits_scores.empty()) {  // search_field.rs:379-383
                std::vector<uint32_t> ids;
                for (auto& h : l.hits_scores) ids.push_back(h.first);
                term_id_hits[l.path][part.terms[0]] = ids;
            }
            if (part.token_value) apply_token_value(*part.token_value, l);  // :391-395
            l.computed = true;
        }
        return l;
    }

    // ------------------------------------------------------------ lists
    uint32_t add_list(const HList& h) {
        if (cq.lists.size() >= size_t(kMaxLists)) unsupported("more than " + std::to_string(kMaxLists) + " posting/id lists in one query");
        cq.lists.push_back(h);
        cq.total_len += h.len;
        return uint32_t(cq.lists.size() - 1);
    }
    uint32_t add_inline_list(std::vector<uint32_t> docs) {  // sorted unique doc ids, restricted to the shard
        std::vector<uint32_t> d;
        for (uint32_t x : docs)
            if (x >= idx.doc_lo && x < idx.doc_hi) d.push_back(x);
        HList h;
        h.len = uint32_t(d.size());
        h.global_len = docs.size();
        h.inline_idx = int(cq.inline_lists.size());
        cq.inline_lists.push_back(std::move(d));
        return add_list(h);
    }
    const PostingStore& posting_store(const Leaf& l) {  // (memoised in the leaf: the lookup builds a path string)
        if (!l.store) const_cast<Leaf&>(l).store = &posting_store(l.path);
        return *l.store;
    }
    const PostingStore& posting_store(const std::string& textindex_path) {
        auto it = idx.postings.find(textindex_path + TO_ANCHOR_ID_SCORE);
        if (it == idx.postings.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + textindex_path + TO_ANCHOR_ID_SCORE);
        return it->second;
    }
    const KVStore& kv_store(const std::string& path) {
        auto it = idx.kv.find(path);
        if (it == idx.kv.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
        return it->second;
    }

    // hits_ids of resolve_token_to_anchor (search_field.rs:468-498): text ids -> anchors, as id-only lists.
    // Returns the list indices; `with_multiplicity` keeps duplicates as extra lists (boost_term).
    std::vector<uint32_t> ids_to_anchor_lists(const Leaf& l, bool with_multiplicity) {
        std::vector<uint32_t> out;
        if (l.hits_ids.empty()) return out;
        if (idx.is_anchor_identity(l.path)) {  // text ids ARE anchor ids
            std::vector<uint32_t> ids = l.hits_ids;
            std::sort(ids.begin(), ids.end());
            if (!with_multiplicity) ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            out = layered_inline(ids);
            return out;
        }
        const KVStore& t2a = kv_store(l.path + TEXT_ID_TO_ANCHOR);
        if ((!with_multiplicity && l.hits_ids.size() <= union_min()) || (l.hits_ids.size() == 1 && t2a.rows_sorted_unique)) {
            for (uint32_t id : l.hits_ids) {
                if (id < t2a.key_base || id - t2a.key_base >= t2a.num_keys) continue;
                const uint32_t r = id - t2a.key_base;
                if (t2a.host_off[r] == t2a.host_off[r + 1]) continue;
                HList h;
                h.d_docs = t2a.values.as<uint32_t>() + t2a.start[r];
                h.len = t2a.len[r];
                h.global_len = t2a.host_off[r + 1] - t2a.host_off[r];
                out.push_back(add_list(h));
            }
            return out;
        }
        std::vector<uint32_t> all;
        for (uint32_t id : l.hits_ids) {
            const uint32_t *b, *e;
            if (t2a.host_row(id, &b, &e)) all.insert(all.end(), b, e);
        }
        std::sort(all.begin(), all.end());
        if (!with_multiplicity) all.erase(std::unique(all.begin(), all.end()), all.end());  // wide expansion: united on the host
        return layered_inline(all);
    }
    // sorted ids with duplicates -> layers of unique lists (layer j holds the ids occurring more than j times)
    std::vector<uint32_t> layered_inline(const std::vector<uint32_t>& sorted_ids) {
        std::vector<std::vector<uint32_t>> layers;
        for (size_t i = 0; i < sorted_ids.size();) {
            size_t j = i;
            while (j < sorted_ids.size() && sorted_ids[j] == sorted_ids[i]) ++j;
            const size_t mult = j - i;
            if (layers.size() < mult) layers.resize(mult);
            for (size_t m = 0; m < mult; ++m) layers[m].push_back(sorted_ids[i]);
            i = j;
        }
        std::vector<uint32_t> out;
        for (auto& l : layers) out.push_back(add_inline_list(std::move(l)));
        return out;
    }

    static size_t union_min() {  // leaves with more posting lists than this are materialised by k_union first
        static const size_t v = std::getenv("VQ_UNION_MIN") ? size_t(std::atoi(std::getenv("VQ_UNION_MIN"))) : 4;
        return v;
    }
    static bool has_wide_and(const SearchRequest& r) {
        if (r.kind == SearchRequest::Search) return false;
        if (r.kind == SearchRequest::And && r.tree.queries.size() >= 3) return true;
        for (auto& q : r.tree.queries)
            if (has_wide_and(q)) return true;
        return false;
    }

    // ------------------------------------------------------------ trees
    void push_op(std::vector<DOp>& ops, const DOp& op, uint32_t& sp) {
        if (ops.size() >= size_t(kMaxOps)) unsupported("query tree with more than " + std::to_string(kMaxOps) + " nodes");
        ops.push_back(op);
        if (op.kind != OP_LEAF) sp -= op.nchild;
        ++sp;
        max_depth = std::max(max_depth, sp);
        if (sp > uint32_t(kStackDepth)) unsupported("query tree deeper than the evaluation stack");
    }

    // key of the union job that materialises this leaf, or "" when the leaf's lists are scanned as they are (same rule and key as
    // in compile_leaf_lists)
    std::string leaf_union_key(const Leaf& l) {
        const PostingStore& ps = posting_store(l);
        size_t nonempty = 0;
        for (auto& [tid, score] : l.hits_scores)
            if (tid < ps.num_tokens && ps.global_len[tid]) ++nonempty;
        if (!(nonempty > union_min() || (nonempty > 1 && req.search_req && has_wide_and(*req.search_req)))) return std::string();
        // (the matched terms and their scores follow from the leaf's request alone: its key identifies the merged list)
        return "u|" + l.path + "|" + (l.key.empty() ? l.part->key() : l.key);
    }

    // score leaf: the posting lists of the matched terms (resolve_token_to_anchor, search_field.rs:400-504)
    NodeInfo compile_leaf_scores(const SearchRequest& r, Leaf& l, std::vector<DOp>& ops, uint32_t& sp) {
        std::vector<Leaf*> one{&l};
        return compile_leaf_lists(r.part.terms[0], one, ops, sp);
    }

    // One OP_LEAF over the posting lists of `members`' matched terms: the union of the lists, per doc the largest s_t * (f16 / 100).
    // One member: a leaf of the request (search_field.rs:453-464 keeps the maximum per doc).  Several: leaves of ONE OR node that carry the
    // same term — the query generator's expansion of a term over the fields — fused: they share a term slot there, and a slot's value is the
    // maximum over its operands (set_op.rs:169-177), i.e. the same maximum over the same values; hits and scores are unchanged, the node keeps
    // one operand per distinct term.
    NodeInfo compile_leaf_lists(const std::string& label, const std::vector<Leaf*>& members, std::vector<DOp>& ops, uint32_t& sp) {
        PhaseTimer pt(3);
        NodeInfo info;
        DOp op{};
        op.kind = OP_LEAF;
        op.list_begin = uint16_t(cq.lists.size());
        info.label = label;
        struct Entry {
            const PostingStore* ps;
            uint32_t tid;
            float score;
        };
        std::vector<Entry> entries;
        entries.reserve(4);
        info.cover.reserve(4);
        for (Leaf* l : members) {
            const PostingStore& ps = posting_store(*l);
            for (auto& [tid, score] : l->hits_scores)
                if (tid < ps.num_tokens) entries.push_back({&ps, tid, score});
        }
        size_t nonempty = 0;  // lists with entries anywhere: every shard must take the same decision (the pre-passes are collective)
        for (auto& e : entries)
            if (e.ps->global_len[e.tid]) ++nonempty;
        uint32_t count = 0;
        if (nonempty > union_min() || (nonempty > 1 && req.search_req && has_wide_and(*req.search_req))) {
            // K2: the leaf's hits are materialised once per batch (union, max per doc) and scanned as one list
            std::string ukey;
            if (members.size() == 1) ukey = leaf_union_key(*members[0]);
            else {
                ukey = "u|fused";
                for (Leaf* l : members) ukey += "|" + l->path + "|" + (l->key.empty() ? l->part->key() : l->key);
            }
            const UnionJob* done = nullptr;
            if (unions) {
                auto it = unions->find(ukey);
                if (it != unions->end()) done = &it->second;
            }
            info.glen = 0;
            if (!done) {
                UnionJob job;
                job.key = ukey;
                for (auto& e : entries)
                    if (e.ps->global_len[e.tid]) {
                        job.terms.push_back(UnionJob::Term{e.ps, e.tid, e.score});
                        job.input_postings += e.ps->len[e.tid];
                    }
                cq.union_requests.push_back(std::move(job));  // compiled again after the jobs ran
                info.len_known = true;
            } else {
                if (done->len) {
                    HList h;
                    h.d_docs = done->d_docs;
                    h.d_scores = reinterpret_cast<const uint16_t*>(done->d_vals);
                    h.len = done->len;
                    h.global_len = done->global_len;
                    h.flags = LIST_HAS_SCORES | LIST_F32;
                    h.term_score = 1.0f;
                    h.max_value = done->max_value;
                    uint32_t li = add_list(h);
                    info.cover.push_back(li);
                    info.cover_len += h.len;
                    count = 1;
                }
                cq.algorithmic_bytes += 6ull * done->input_postings + 8ull * done->len;
                info.glen = done->global_len;
                // the merged length is this shard's only: known globally when the index is not sharded
                info.len_known = !req.filter && idx.can_sum_over_shards();
            }
            op.list_count = uint16_t(count);
            push_op(ops, op, sp);
            info.emitted = true;
            return info;
        }
        uint32_t with_entries = 0;
        for (auto& e : entries) {
            const PostingStore& ps = *e.ps;
            if (ps.global_len[e.tid]) ++with_entries;
            if (ps.len[e.tid] == 0) continue;
            HList h;
            h.d_docs = ps.docs.as<uint32_t>() + ps.start[e.tid];
            h.d_scores = ps.scores.as<uint16_t>() + ps.start[e.tid];
            h.len = ps.len[e.tid];
            h.global_len = ps.global_len[e.tid];
            h.flags = LIST_HAS_SCORES;
            h.term_score = e.score;
            h.max_raw = ps.max_raw[e.tid];
            if (!ps.bm_start.empty() && ps.bm_start[e.tid] >= 0) {
                h.flags |= LIST_BITMAP;
                h.d_bitmap = ps.bitmaps.as<uint32_t>() + ps.bm_start[e.tid];
                h.d_rank_dir = ps.rank_dir.as<uint32_t>() + ps.rd_start[e.tid];
            }
            if (!ps.td_start.empty() && ps.td_start[e.tid] >= 0) h.d_tile_dir = ps.tile_dir.as<uint32_t>() + ps.td_start[e.tid];
            if (!ps.pk_start.empty() && ps.pk_start[e.tid] >= 0) {
                h.d_cov32 = ps.cov32.as<uint32_t>() + uint64_t(ps.pk_start[e.tid]) * 8;
                h.d_gdir = ps.gdir.as<uint32_t>() + ps.gd_start[e.tid];
                if (ps.ak_start[e.tid] >= 0) h.d_arr16 = ps.arr16.as<uint16_t>() + uint64_t(ps.ak_start[e.tid]) * 8;
            }
            uint32_t li = add_list(h);
            info.cover.push_back(li);
            info.cover_len += h.len;
            cq.algorithmic_bytes += 6ull * h.len;
            ++count;
        }
        op.list_count = uint16_t(count);
        // a single posting list and no Set filter in front of it: the result length is the list length
        info.len_known = with_entries <= 1 && !req.filter;
        info.glen = 0;
        for (auto& e : entries) info.glen += e.ps->global_len[e.tid];
        push_op(ops, op, sp);
        info.emitted = true;
        return info;
    }

    void fill_boost_params(DColBoost& cb, const RequestBoostPart& b) {
        cb.fun = b.boost_fun ? int32_t(*b.boost_fun) : BF_NONE;  // enum order matches BoostFun
        cb.param = b.param.value_or(0.0f);
        if (b.skip_when_score) {
            if (b.skip_when_score->size() > size_t(kMaxSkipWhen)) unsupported("more than 4 skip_when_score values");
            cb.nskip = uint32_t(b.skip_when_score->size());
            for (size_t i = 0; i < b.skip_when_score->size(); ++i) cb.skip[i] = (*b.skip_when_score)[i];
        }
        cb.expr_op = EX_NONE;
        if (b.expression) parse_expression(*b.expression, cb);
    }

    // matched terms -> text ids -> value ids of the 1:n object -> (anchor, boost value) of every boosted value id (boost.rs:432-468)
    // the text ids a leaf's matched terms stand for (resolve_token_hits_to_text_id_ids_only, search_field.rs:640-689)
    std::vector<uint32_t> boost_1n_text_ids(const RequestSearchPart& part, Leaf& l) {
        std::vector<uint32_t> ids;
        auto cit = idx.columns.find(part.path);
        const bool tokenized = cit != idx.columns.end() && cit->second.tokenize;
        if (tokenized) {
            const KVStore& t2t = kv_store(l.path + TOKENS_TO_TEXT_ID);
            for (auto& h : l.hits_scores) {
                const uint32_t *rb, *re;
                if (t2t.host_row(h.first, &rb, &re)) ids.insert(ids.end(), rb, re);
                else ids.push_back(h.first);  // is a text id
            }
            sort_unique_u32(ids);
        } else ids = l.hits_ids;
        return ids;
    }
    std::vector<std::pair<uint32_t, float>> resolve_boost_1n(const RequestSearchPart& part, Leaf& l, const RequestBoostPart& b) {
        PhaseTimer pt(1);
        std::vector<uint32_t> value_ids;  // join_to_parent_ids search.rs:281-315
        {
            PhaseTimer gather(6);
            const std::vector<uint32_t> ids = boost_1n_text_ids(part, l);
            const KVStore& to_parent = kv_store(l.path + VALUE_ID_TO_PARENT);
            for (uint32_t id : ids) {
                const uint32_t *rb, *re;
                if (to_parent.host_row(id, &rb, &re)) value_ids.insert(value_ids.end(), rb, re);
            }
        }
        {
            PhaseTimer sort(7);
            sort_unique_u32(value_ids);
        }
        PhaseTimer pairs_timer(8);
        auto bit = idx.boost.find(b.path + BOOST_VALID_TO_VALUE);
        if (bit == idx.boost.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + b.path + BOOST_VALID_TO_VALUE);
        const KVStore& to_anchor = kv_store(b.path + VALUE_ID_TO_ANCHOR);
        std::vector<std::pair<uint32_t, float>> pairs;  // (anchor, boost value) in value-id order
        pairs.reserve(value_ids.size());
        for (uint32_t vid : value_ids) {
            float v;
            if (!bit->second.host_value(vid, &v)) continue;
            const uint32_t *rb, *re;
            if (to_anchor.host_row(vid, &rb, &re)) pairs.push_back({*rb, v});
        }
        return pairs;
    }

    // BoostToAnchor + ApplyAnchorBoost (plan_steps.rs:174-219): matched terms -> text ids -> value ids of the 1:n object ->
    // boost value and anchor of each value id (boost.rs:432-468), applied to the leaf's hits by anchor (boost.rs:255-281).
    void emit_boost_1n(const RequestSearchPart& part, Leaf& l, const RequestBoostPart& b, std::vector<DOp>& ops, uint32_t& sp) {
        PhaseTimer pt(2);
        // Everything below that is proportional to the boost list — resolving it, checking its order, cutting it into layers — is done ONCE per
        // (leaf request, boost path) and batch: by whichever request gets there first, for all compilation passes and all requests that share
        // the leaf (Boost1nCache).  What remains per request is copying the layers' arrays into its blob.
        const std::string cache_key = (l.key.empty() ? part.key() : l.key) + "|" + b.path;
        auto emit_layer = [&](const HList& h) {
            const uint32_t li = add_list(h);
            cq.algorithmic_bytes += 8ull * h.len;
            DColBoost cb{};
            fill_boost_params(cb, b);
            cb.nskip = 0;  // apply_boost_values_anchor has no skip_when_score
            cq.leaf_cols.push_back(cb);
            DOp op{};
            op.kind = OP_BOOST1N;
            op.nchild = 1;
            op.list_begin = uint16_t(li);
            op.list_count = 1;
            op.child_slot[0] = uint8_t(cq.leaf_cols.size() - 1);  // rebased behind the request-level boosts when the query is finished
            push_op(ops, op, sp);
        };
        // K10 (VQ_BOOST1N_DEVICE=1): the list is resolved on the device (gather of the value ids, sort, boost value and anchor of each) — the compiler
        // only learns its length and whether an anchor carries several values; then the look-ahead rule needs the pairs themselves: host path below.
        // Off by default: measured on the reference's bench_jmdict request it moves 4 ms of host work per 256 requests into a 2.4 ms pre-pass
        // (segmented sort 1.2 ms, one wave per list 1.1 ms) plus a third compilation pass, and the longest lists — prefix matches, which are
        // the ones with several values per anchor — end on the host path anyway: 14.3 k instead of 17.3 k requests/s.
        static const bool device_on = std::getenv("VQ_BOOST1N_DEVICE") != nullptr;
        if (boost_cache && device_on) {
            const std::string to_parent = l.path + VALUE_ID_TO_PARENT, to_anchor = b.path + VALUE_ID_TO_ANCHOR, col = b.path + BOOST_VALID_TO_VALUE;
            auto kp = idx.kv.find(to_parent), ka = idx.kv.find(to_anchor);
            const bool staged = kp != idx.kv.end() && ka != idx.kv.end() && kp->second.value_csr && ka->second.value_csr && idx.boost.count(col);
            if (staged && !boost_cache->device) {  // first pass: ask for the list
                Boost1nJob job;
                job.key = cache_key;
                job.to_parent_path = to_parent;
                job.to_anchor_path = to_anchor;
                job.boost_path = col;
                job.text_ids = boost_1n_text_ids(part, l);
                cq.boost1n_requests.push_back(std::move(job));
                return;  // (this compilation is thrown away)
            }
            if (staged) {
                auto it = boost_cache->device->find(cache_key);
                if (it != boost_cache->device->end() && it->second.done) {
                    const Boost1nJob& job = it->second;
                    if (!job.ascending) unsupported("1:n field boost whose value ids are not in anchor order (" + b.path + ")");
                    if (!job.several) {
                        HList h;
                        h.d_docs = job.d_docs;
                        h.d_scores = reinterpret_cast<const uint16_t*>(job.d_vals);
                        h.len = job.len;
                        h.global_len = job.total;
                        h.flags = LIST_HAS_SCORES | LIST_F32;
                        h.term_score = 1.0f;
                        emit_layer(h);
                        return;
                    }
                }
            }
        }
        std::shared_ptr<Boost1nEntry> entry = boost_cache ? boost_cache->entry(cache_key) : std::make_shared<Boost1nEntry>();
        auto make_layers = [&](const std::vector<std::vector<std::pair<uint32_t, float>>>& layers) {
            auto out = std::make_shared<std::vector<Boost1nEntry::Layer>>();
            for (auto& layer : layers) {
                Boost1nEntry::Layer L;
                L.global_len = layer.size();
                L.docs.reserve(layer.size());
                L.vals.reserve(layer.size());
                for (auto& pr : layer)
                    if (pr.first >= idx.doc_lo && pr.first < idx.doc_hi) {
                        L.docs.push_back(pr.first);
                        L.vals.push_back(pr.second);
                    }
                out->push_back(std::move(L));
            }
            return out;
        };
        std::call_once(entry->resolved, [&] {
            entry->pairs = resolve_boost_1n(part, l, b);  // (anchor, boost value) in value-id order
            PhaseTimer order_check(9);
            const auto& pairs = entry->pairs;
            for (size_t i = 1; i < pairs.size(); ++i) {
                if (pairs[i].first < pairs[i - 1].first) entry->ascending = false;
                entry->several = entry->several || pairs[i].first == pairs[i - 1].first;
            }
            if (!entry->ascending) return;
            if (!entry->several) entry->layers = make_layers({pairs});
            else {
                entry->anchors.reserve(pairs.size());
                for (auto& pr : pairs)
                    if (entry->anchors.empty() || entry->anchors.back() != pr.first) entry->anchors.push_back(pr.first);
            }
        });
        if (!entry->ascending) unsupported("1:n field boost whose value ids are not in anchor order (" + b.path + ")");
        std::shared_ptr<const std::vector<Boost1nEntry::Layer>> layers = entry->layers;
        if (entry->several) {
            // Several boosted values on one anchor.  The reference walks the boost list against the leaf's hits with one look-ahead
            // entry (boost.rs:262-279): an anchor reached while the look-ahead already rests on its first entry gets that entry ONLY
            // (the rest is skipped at the next hit); reached by scanning forward, it gets ALL its entries.  With the entry anchors
            // a_0 < a_1 < ...: a_j is met resting on its first entry iff it is the very first entry, or the leaf has a hit strictly
            // between a_(j-1) and a_j, or a_(j-1) is a hit that was NOT met that way (its scan stopped on a_j's first entry).
            if (req.filter) unsupported("1:n field boost with several boosted values on one anchor, under a filter (" + b.path + ")");
            const std::vector<uint32_t>& anchors = entry->anchors;
            const auto& pairs = entry->pairs;
            const RangeJob* done = nullptr;
            if (ranges) {
                auto it = ranges->find(cache_key);
                if (it != ranges->end()) done = &it->second;
            }
            if (!done) {  // ask for the leaf's hits at every entry anchor and between neighbouring ones; compiled again afterwards
                RangeJob job;
                job.key = cache_key;
                job.store_path = l.path + TO_ANCHOR_ID_SCORE;
                job.union_key = leaf_union_key(l);
                job.tokens.reserve(l.hits_scores.size());
                for (auto& h : l.hits_scores) job.tokens.push_back(h.first);
                job.anchors = std::shared_ptr<const std::vector<uint32_t>>(entry, &entry->anchors);
                cq.range_requests.push_back(std::move(job));
                // (placeholder: this compilation is thrown away)
                static const auto empty = std::make_shared<const std::vector<Boost1nEntry::Layer>>(1);
                layers = empty;
            } else {
                if (done->counts.size() != 2 * anchors.size()) unsupported("1:n field boost: range pre-pass does not match the boost list (internal)");
                std::call_once(entry->layered, [&] {
                    std::vector<std::vector<std::pair<uint32_t, float>>> cut(1);
                    bool prev_hit = false, prev_first_only = false;
                    size_t p = 0;
                    for (size_t j = 0; j < anchors.size(); ++j) {
                        const bool hit = done->counts[2 * j] != 0, between = j && done->counts[2 * j + 1] != 0;
                        const bool first_only = j == 0 || between || (prev_hit && !prev_first_only);
                        size_t e = p;
                        while (e < pairs.size() && pairs[e].first == anchors[j]) ++e;
                        const size_t take = first_only ? 1 : e - p;
                        if (hit)
                            for (size_t r = 0; r < take; ++r) {
                                if (cut.size() <= r) cut.emplace_back();
                                cut[r].push_back(pairs[p + r]);
                            }
                        p = e;
                        prev_hit = hit;
                        prev_first_only = first_only;
                    }
                    entry->layers = make_layers(cut);
                });
                layers = entry->layers;
                if (layers->size() > 8) unsupported("1:n field boost with more than 8 boosted values on one anchor (" + b.path + ")");
            }
        }
        for (const Boost1nEntry::Layer& layer : *layers) {
            HList h;
            h.len = uint32_t(layer.docs.size());
            h.global_len = layer.global_len;
            h.flags = LIST_HAS_SCORES | LIST_F32;
            h.term_score = 1.0f;
            h.inline_idx = int(cq.inline_lists.size());
            h.inline_val_idx = int(cq.inline_vals.size());
            cq.inline_lists.push_back(layer.docs);
            cq.inline_vals.push_back(layer.vals);
            emit_layer(h);
        }
    }


    NodeInfo compile_node(const SearchRequest& r, bool is_filter, std::vector<DOp>& ops, uint32_t& sp, const std::vector<RequestBoostPart>& boost) {
        NodeInfo info = compile_node_inner(r, is_filter, ops, sp, boost, is_filter ? UINT32_MAX : next_node++);
        return info;
    }
    // result length of a node as the reference sees it: measured by the count pre-pass when it is not known statically
    void apply_counts(NodeInfo& c) {
        if (c.len_known || !counts) return;
        auto it = counts->nodes.find(c.node_id);
        if (it == counts->nodes.end()) return;
        const bool filter_is_set = counts->has_filter && counts->filter_count <= 100000;  // FilterResult::Set: leaves drop ids outside it
        c.glen = filter_is_set ? it->second.second : it->second.first;
        c.len_known = true;
    }
    // the 1:n boost joined to a search path through a shared [] prefix (execution_plan.rs:422-509)
    const RequestBoostPart* find_boost_1n(const RequestSearchPart& part, const std::vector<RequestBoostPart>& boost) {
        const RequestBoostPart* boost_1n = nullptr;
        size_t pos = part.path.rfind("[]");
        if (pos == std::string::npos) return nullptr;
        std::string end_obj = part.path.substr(0, pos);
        for (auto& el : boost) {
            size_t p = el.path.rfind("[]");
            if (p != std::string::npos && el.path.substr(0, p) == end_obj) {
                if (boost_1n) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"more than one boost matches the 1:n search path\" ");
                boost_1n = &el;
            }
        }
        return boost_1n;
    }
    // bookkeeping of a score leaf's result that does not depend on how its lists are scanned
    void note_score_leaf(Leaf& l) {
        if (l.store_term_texts && !l.terms.empty()) {  // search_field.rs:386-389, merged upwards by set_op.rs:49-63 (a leaf used twice counts twice)
            auto& dst = cq.why_found_terms[l.path];
            for (auto& t : l.terms) dst.push_back(t.second);
        }
    }
    NodeInfo compile_node_inner(const SearchRequest& r, bool is_filter, std::vector<DOp>& ops, uint32_t& sp, const std::vector<RequestBoostPart>& boost,
                                uint32_t my_id) {
        NodeInfo info;
        info.node_id = my_id;
        if (r.kind == SearchRequest::Search) {
            Leaf& l = field_result(r.part);
            if (!is_filter) note_score_leaf(l);
            const RequestBoostPart* boost_1n = is_filter ? nullptr : find_boost_1n(r.part, boost);
            if (boost_1n && explain_on)  // join_to_parent_ids looks the value ids' records up under the TEXT ids (search.rs:301): a panic unless the ids coincide
                unsupported("explain with a 1:n boost (the reference panics in join_to_parent_ids, search.rs:301)");
            if (boost_1n) {  // the leaf, then a unary op that applies the per-anchor boost values
                NodeInfo leaf_info = compile_leaf_scores(r, l, ops, sp);
                leaf_info.node_id = my_id;
                leaf_info.root_op = int(ops.size()) - 1;
                emit_boost_1n(r.part, l, *boost_1n, ops, sp);
                leaf_info.emitted = true;
                return leaf_info;
            }
            if (!is_filter) {
                NodeInfo leaf_info = compile_leaf_scores(r, l, ops, sp);
                leaf_info.node_id = my_id;
                leaf_info.root_op = int(ops.size()) - 1;
                if (explain_on) leaf_info.ex_nodes = {ex_leaf(&l)};
                return leaf_info;
            }
            DOp op{};
            op.kind = OP_LEAF;
            op.list_begin = uint16_t(cq.lists.size());
            info.label = r.part.terms[0];
            if (is_filter) {
                auto lists = ids_to_anchor_lists(l, false);
                op.list_count = uint16_t(lists.size());
                for (uint32_t li : lists) {
                    info.cover.push_back(li);
                    info.cover_len += cq.lists[li].len;
                    info.glen += cq.lists[li].global_len;
                    cq.algorithmic_bytes += 4ull * cq.lists[li].len;
                }
                info.len_known = lists.size() <= 1;
            }
            push_op(ops, op, sp);
            info.emitted = true;
            return info;
        }
        const auto& queries = r.tree.queries;
        if (queries.empty()) {  // set_op.rs:90-92 / :369-371: empty result
            DOp op{};
            op.kind = OP_LEAF;
            op.list_begin = uint16_t(cq.lists.size());
            op.list_count = 0;
            push_op(ops, op, sp);
            info.len_known = true;
            info.label_known = false;
            info.emitted = true;
            info.root_op = int(ops.size()) - 1;
            if (explain_on && !is_filter) info.ex_nodes = {ex_leaf(nullptr)};
            return info;
        }
        // Operands of a scored OR that are plain leaves with the same term — the query generator's expansion of one term over the searched
        // fields (query_parser_to_veloci_request.rs:84-109) — share a term slot (set_op.rs:143), and a slot's value is the maximum over its
        // operands (:169-177): they are compiled as ONE leaf over all their posting lists (compile_leaf_lists), the OR keeps one operand per
        // distinct term.  fuse_with[i]: the operands fused into operand i (itself first); skip[i]: operand i was fused into an earlier one.
        static const bool no_fuse = std::getenv("VQ_NO_LEAF_FUSION") != nullptr;
        std::vector<std::vector<size_t>> fuse_with(queries.size());
        std::vector<bool> skip(queries.size(), false);
        if (!is_filter && r.kind == SearchRequest::Or && queries.size() > 1 && !no_fuse) {
            std::map<std::string, size_t> first_of;
            for (size_t i = 0; i < queries.size(); ++i) {
                const SearchRequest& q = queries[i];
                const bool own_options = q.part.options && (q.part.options->boost || q.part.options->top || q.part.options->skip);  // (explain alone changes nothing here)
                if (q.kind != SearchRequest::Search || q.part.terms.empty() || own_options || find_boost_1n(q.part, boost)) continue;
                auto [it, fresh] = first_of.emplace(q.part.terms[0], i);
                fuse_with[it->second].push_back(i);
                skip[i] = !fresh;
            }
        }
        std::vector<NodeInfo> ch;
        ch.reserve(queries.size());
        for (size_t qi = 0; qi < queries.size(); ++qi) {
            const SearchRequest& q = queries[qi];
            if (skip[qi]) continue;
            if (fuse_with[qi].size() > 1) {
                std::vector<Leaf*> members;
                uint32_t first_id = UINT32_MAX;
                for (size_t m : fuse_with[qi]) {
                    const uint32_t id = next_node++;
                    if (first_id == UINT32_MAX) first_id = id;
                    Leaf& l = field_result(queries[m].part);
                    note_score_leaf(l);
                    members.push_back(&l);
                }
                NodeInfo fused = compile_leaf_lists(q.part.terms[0], members, ops, sp);
                if (explain_on)
                    for (size_t k = 0; k < members.size(); ++k) {
                        fused.ex_nodes.push_back(ex_leaf(members[k]));
                        fused.ex_pos.push_back(fuse_with[qi][k]);
                    }
                fused.node_id = first_id;
                fused.root_op = int(ops.size()) - 1;
                ch.push_back(std::move(fused));
                continue;
            }
            std::vector<RequestBoostPart> child_boost = boost;  // merge_vec execution_plan.rs:263-270
            if (q.get_options() && q.get_options()->boost) child_boost.insert(child_boost.end(), q.get_options()->boost->begin(), q.get_options()->boost->end());
            ch.push_back(compile_node(q, is_filter, ops, sp, child_boost));
            if (explain_on && !is_filter) ch.back().ex_pos = {qi};
            if (is_filter && ch.size() >= 2) {
                // presence only: AND/OR are associative, fold pairwise so the stack stays shallow
                DOp op{};
                op.kind = r.kind == SearchRequest::And ? OP_AND : OP_OR;
                op.nchild = 2;
                push_op(ops, op, sp);
            }
        }
        if (queries.size() == 1) return ch[0];  // set_op.rs:93-96 / :372-375: the single operand is passed through
        if (is_filter) {
            info.label_known = false;
            info.emitted = true;
            if (r.kind == SearchRequest::And) {
                size_t best = 0;
                for (size_t i = 1; i < ch.size(); ++i)
                    if (ch[i].cover_len < ch[best].cover_len) best = i;
                info.cover = ch[best].cover;
                info.cover_len = ch[best].cover_len;
            } else {
                for (auto& c : ch) {
                    info.cover.insert(info.cover.end(), c.cover.begin(), c.cover.end());
                    info.cover_len += c.cover_len;
                }
            }
            return info;
        }
        if (ch.size() > size_t(kMaxChildren)) unsupported("more than " + std::to_string(kMaxChildren) + " operands in one and/or");
        DOp op{};
        op.nchild = uint8_t(ch.size());
        const size_t n = ch.size();
        if (r.kind == SearchRequest::And) {
            op.kind = OP_AND;
            bool all_known = true;
            for (auto& c : ch) {
                apply_counts(c);
                all_known = all_known && c.len_known;
            }
            size_t shortest = 0;
            if (all_known) {  // get_shortest_result set_op.rs:9-17: first minimal length
                for (size_t i = 1; i < n; ++i)
                    if (ch[i].glen < ch[shortest].glen) shortest = i;
            } else {
                // The summation order (set_op.rs:388-416) and the label of the result (:439) follow the operands' result sizes, which only
                // exist at run time here: ask for a count pre-pass (presence only) and compile again with its numbers.
                // (two operands: the sum of two floats does not depend on the order; only the label does, and an OR that needs it says so)
                // (explain: the records of the shortest operand are the ones left out, :393,:421-433 — its identity matters for two operands as well)
                const bool sizes_needed = n > 2 || explain_on;
                if (counts && sizes_needed) unsupported("AND whose operand sizes are still unknown after the count pre-pass (internal)");
                if (!counts) {
                    if (sizes_needed && !idx.can_sum_over_shards())
                        unsupported("AND of 3+ operands whose result sizes are only known at run time, on a sharded index without vq_index_set_allreduce");
                    auto& dst = sizes_needed ? count_reqs : maybe_reqs;
                    for (auto& c : ch)
                        if (!c.len_known) dst.push_back({c.node_id, c.root_op});
                }
            }
            // swap_remove(shortest): the last operand takes its slot; the shortest is added last (:393,:415-416)
            std::vector<uint8_t> order;
            order.reserve(n);
            for (size_t i = 0; i < n; ++i) order.push_back(uint8_t(i));
            order[shortest] = order[n - 1];
            order.pop_back();
            order.push_back(uint8_t(shortest));
            for (size_t i = 0; i < n; ++i) op.and_order[i] = order[i];
            // result.request = and_results[0].request after the swap_remove (:439)
            const size_t label_src = order[0];
            info.label = ch[label_src].label;
            info.label_known = (all_known || n == 1) && ch[label_src].label_known;
            if (!all_known && n == 2) info.label_known = false;
            info.len_known = false;
            // cover: the operand with the fewest shard-local entries
            size_t best = 0;
            for (size_t i = 1; i < n; ++i)
                if (ch[i].cover_len < ch[best].cover_len) best = i;
            info.cover = ch[best].cover;
            info.cover_len = ch[best].cover_len;
        } else {
            op.kind = OP_OR;
            if (!is_filter) {
                std::vector<std::string> terms;  // set_op.rs:122-124
                for (auto& c : ch) {
                    if (!c.label_known) {  // label of a nested AND: known once its operands' sizes are (count pre-pass)
                        if (counts) unsupported("OR over an operand whose term label is only known at run time (set_op.rs:143,439)");
                        label_wanted = true;
                    }
                    terms.push_back(c.label);
                }
                std::sort(terms.begin(), terms.end());
                terms.erase(std::unique(terms.begin(), terms.end()), terms.end());
                op.nslots = uint8_t(terms.size());
                for (size_t i = 0; i < n; ++i) op.child_slot[i] = uint8_t(std::find(terms.begin(), terms.end(), ch[i].label) - terms.begin());
            }
            info.label = ch[0].label;  // set_op.rs:215
            info.label_known = ch[0].label_known;
            info.len_known = false;
            for (auto& c : ch) {
                info.cover.insert(info.cover.end(), c.cover.begin(), c.cover.end());
                info.cover_len += c.cover_len;
            }
        }
        push_op(ops, op, sp);
        info.emitted = true;
        info.root_op = int(ops.size()) - 1;
        if (explain_on) {
            ExplainNode xn;
            xn.kind = r.kind == SearchRequest::And ? XP_AND : XP_OR;
            if (r.kind == SearchRequest::And) {
                for (size_t i = 0; i < n; ++i) {
                    xn.children.push_back(ch[i].ex_nodes.at(0));
                    xn.order.push_back(op.and_order[i]);
                }
            } else {  // the operands in request order, every member of a fused leaf with the leaf's term slot
                std::vector<std::tuple<size_t, int, uint16_t>> operands;
                for (size_t i = 0; i < n; ++i)
                    for (size_t k = 0; k < ch[i].ex_nodes.size(); ++k) operands.emplace_back(ch[i].ex_pos.at(k), ch[i].ex_nodes[k], uint16_t(op.child_slot[i]));
                std::sort(operands.begin(), operands.end());
                for (auto& [pos, node, slot] : operands) {
                    xn.children.push_back(node);
                    xn.order.push_back(slot);
                }
            }
            xplan->nodes.push_back(std::move(xn));
            info.ex_nodes = {int(xplan->nodes.size()) - 1};
        }
        return info;
    }

    // Presence program for k_tile_scan P3: simulate the postfix stacks of the score tree and the filter tree on
    // slot references and emit three-address ops; the last op writes the root words (score root AND filter root).
    void build_presence_program() {
        std::vector<uint16_t> free_temps;
        uint32_t next_temp = 0;
        auto alloc_temp = [&]() -> uint16_t {
            if (!free_temps.empty()) {
                uint16_t t = free_temps.back();
                free_temps.pop_back();
                return t;
            }
            return uint16_t(kSlotTemp | next_temp++);
        };
        auto release = [&](uint16_t ref) {
            if (ref & kSlotTemp) free_temps.push_back(ref);
        };
        auto emit = [&](uint8_t kind, const std::vector<uint16_t>& in, uint16_t out) {
            DPresOp op{};
            op.kind = kind;
            op.out = out;
            op.n_in = uint16_t(in.size());
            op.in_begin = uint16_t(cq.pres_in.size());
            cq.pres_in.insert(cq.pres_in.end(), in.begin(), in.end());
            cq.pres.push_back(op);
        };
        // count pre-pass: counter 2i = hits of node count_reqs[i], 2i+1 = hits inside the filter, last = ids of the filter
        const bool counting = !count_reqs.empty();
        uint16_t filter_root = 0;
        bool have_filter_root = false;
        auto count_slot = [&](uint16_t slot, size_t req_index) {
            emit(PRES_COUNT, {slot}, uint16_t(2 * req_index));
            if (have_filter_root) {
                uint16_t t = alloc_temp();
                emit(PRES_AND, {slot, filter_root}, t);
                emit(PRES_COUNT, {t}, uint16_t(2 * req_index + 1));
                release(t);
            }
        };
        auto run = [&](const std::vector<DOp>& ops, bool count_here) -> uint16_t {
            std::vector<uint16_t> st;
            for (size_t o = 0; o < ops.size(); ++o) {
                const DOp& op = ops[o];
                struct AfterOp {  // count the node results this op produces (its slot is on top of the stack)
                    const std::vector<CountReq>& reqs;
                    bool on;
                    size_t o;
                    std::vector<uint16_t>& st;
                    decltype(count_slot)& count;
                    ~AfterOp() {
                        if (!on || st.empty()) return;
                        for (size_t i = 0; i < reqs.size(); ++i)
                            if (reqs[i].root_op == int(o)) count(st.back(), i);
                    }
                } after{count_reqs, count_here && counting, o, st, count_slot};
                if (op.kind == OP_BOOST1N) continue;  // changes scores only
                if (op.kind == OP_LEAF) {
                    if (op.list_count == 1) {
                        st.push_back(op.list_begin);
                    } else {
                        std::vector<uint16_t> in;
                        for (uint32_t j = 0; j < op.list_count; ++j) in.push_back(uint16_t(op.list_begin + j));
                        uint16_t t = alloc_temp();
                        emit(op.list_count ? PRES_OR : PRES_ZERO, in, t);
                        st.push_back(t);
                    }
                } else {
                    std::vector<uint16_t> in(st.end() - op.nchild, st.end());
                    st.resize(st.size() - op.nchild);
                    uint16_t t = alloc_temp();  // allocated before the inputs are released: never aliases an input
                    emit(op.kind == OP_AND ? PRES_AND : PRES_OR, in, t);
                    for (uint16_t r : in) release(r);
                    st.push_back(t);
                }
            }
            return st.empty() ? uint16_t(0) : st.back();
        };
        std::vector<uint16_t> roots;
        bool any = !cq.ops.empty();
        if (!cq.fops.empty()) {  // the filter first: the counts of the score tree's nodes look at its root
            filter_root = run(cq.fops, false);
            have_filter_root = true;
            roots.push_back(filter_root);
            if (counting) emit(PRES_COUNT, {filter_root}, uint16_t(2 * count_reqs.size()));
        }
        if (any) roots.push_back(run(cq.ops, true));
        if (!any) emit(PRES_ZERO, {}, kSlotRoot);
        else emit(PRES_AND, roots, kSlotRoot);
        if (counting) {
            for (auto& r : count_reqs) cq.count_nodes.push_back(r.node_id);
            cq.n_counts = uint32_t(2 * count_reqs.size() + 1);
        }
        cq.n_temps = next_temp;
        if (next_temp > 32) unsupported("presence program needs more than 32 temporary bitmaps");
    }

    // Top-k pruning table of k_tile_scan.  G(k) = the largest score a doc can have when at most k of the score tree's leaf lists hold it:
    // every leaf at its list's largest value, a dynamic programme over the tree for the best split of the k lists among the children
    // (OR: every child its own slot — a sum instead of a max — times (present children)^2; AND: every child present), times the largest
    // product of the bitmap-driven multipliers (phrase groups, term boosts, locality).  All of it is monotone, so G bounds the score.
    void compute_prune_table() {
        static const bool off = std::getenv("VQ_NO_PRUNE") != nullptr;
        if (off || !count_reqs.empty() || !cq.cols.empty() || !cq.facets.empty() || cq.ops.empty()) return;
        for (auto& f : cq.locf)
            if (f.list_count == kLocPrecomputed) return;
        std::vector<uint32_t> leaf_lists;
        for (auto& op : cq.ops) {
            if (op.kind == OP_BOOST1N) return;
            if (op.kind == OP_LEAF)
                for (uint32_t j = 0; j < op.list_count; ++j) leaf_lists.push_back(op.list_begin + j);
        }
        std::sort(leaf_lists.begin(), leaf_lists.end());
        if (std::adjacent_find(leaf_lists.begin(), leaf_lists.end()) != leaf_lists.end()) return;  // a list under two leaves would count once but score twice
        const size_t K = leaf_lists.size();
        if (K < 2 || K > 15) return;
        const double NEG = -1.0;  // "cannot be present"
        struct Node {
            std::vector<double> ub;  // ub[k], k = 0..K: best score with at most k lists (NEG: not present)
            size_t need;             // fewest lists that make the node present
        };
        auto monotone = [&](Node& n) {
            for (size_t k = 1; k <= K; ++k) n.ub[k] = std::max(n.ub[k], n.ub[k - 1]);
        };
        std::vector<Node> st;
        for (auto& op : cq.ops) {
            if (op.kind == OP_LEAF) {
                Node n{std::vector<double>(K + 1, NEG), 1};
                double best = NEG;
                for (uint32_t j = 0; j < op.list_count; ++j) {
                    const HList& l = cq.lists[op.list_begin + j];
                    double v;
                    if (l.flags & LIST_F32) v = l.max_value;
                    else if (l.max_raw >= 0x7C00 || l.term_score < 0.0f) return;
                    else {
                        const uint16_t h = l.max_raw;  // finite non-negative f16
                        const int e = (h >> 10) & 31, m = h & 1023;
                        const double a = e ? std::ldexp(1.0 + m / 1024.0, e - 15) : std::ldexp(m / 1024.0, -14);
                        v = double(l.term_score) * (a / 100.0);
                    }
                    if (!(v >= 0.0) || std::isinf(v)) return;
                    best = std::max(best, v);
                }
                if (op.list_count == 0) n.need = K + 1;
                else
                    for (size_t k = 1; k <= K; ++k) n.ub[k] = best;
                st.push_back(std::move(n));
            } else {
                const size_t nc = op.nchild;
                std::vector<Node> ch(st.end() - nc, st.end());
                st.resize(st.size() - nc);
                Node n{std::vector<double>(K + 1, NEG), 0};
                if (op.kind == OP_AND) {
                    std::vector<double> g(K + 1, NEG);
                    g[0] = 0.0;
                    for (auto& c : ch) {
                        std::vector<double> ng(K + 1, NEG);
                        for (size_t j = 0; j <= K; ++j)
                            if (g[j] >= 0.0)
                                for (size_t kc = std::max<size_t>(c.need, 1); j + kc <= K; ++kc)
                                    if (c.ub[kc] >= 0.0) ng[j + kc] = std::max(ng[j + kc], g[j] + c.ub[kc]);
                        g.swap(ng);
                        n.need += c.need;
                    }
                    for (size_t k = 0; k <= K; ++k) n.ub[k] = g[k];
                } else {  // OR
                    std::vector<std::vector<double>> f(nc + 1, std::vector<double>(K + 1, NEG));  // f[t][j]: t children present, j lists used
                    f[0][0] = 0.0;
                    n.need = K + 1;
                    for (auto& c : ch) {
                        auto nf = f;  // child absent
                        for (size_t t = 0; t < nc; ++t)
                            for (size_t j = 0; j <= K; ++j)
                                if (f[t][j] >= 0.0)
                                    for (size_t kc = std::max<size_t>(c.need, 1); j + kc <= K; ++kc)
                                        if (c.ub[kc] >= 0.0) nf[t + 1][j + kc] = std::max(nf[t + 1][j + kc], f[t][j] + c.ub[kc]);
                        f.swap(nf);
                        n.need = std::min(n.need, c.need);
                    }
                    for (size_t t = 1; t <= nc; ++t)
                        for (size_t j = 0; j <= K; ++j)
                            if (f[t][j] >= 0.0) n.ub[j] = std::max(n.ub[j], f[t][j] * double(t) * double(t));
                }
                monotone(n);
                st.push_back(std::move(n));
            }
        }
        if (st.size() != 1) return;
        double mult = 1.00001;  // the kernel's f32 roundings never exceed this margin
        for (auto& g : cq.groups) {
            if (g.mult < 0.0f) return;
            mult *= std::max(1.0, double(g.mult));
        }
        for (auto& t : cq.tboosts) {
            if (t.mult < 0.0f) return;
            mult *= std::max(1.0, double(t.mult));
        }
        if (!cq.locf.empty()) {
            double cmax = 1;
            for (auto& f : cq.locf) cmax = std::max(cmax, double(f.list_count));
            mult *= std::max(1.0, 2.0 * cmax * cmax);
        }
        const Node& root = st[0];
        for (size_t k = 0; k < 16; ++k) {
            const double g = k <= K ? root.ub[k] : root.ub[K];
            float gf = g < 0.0 ? 0.0f : float(g * mult);
            if (double(gf) < g * mult) gf = std::nextafter(gf, std::numeric_limits<float>::infinity());
            uint32_t bits;
            std::memcpy(&bits, &gf, 4);
            cq.prune_gbits[k] = order_f32(bits);
        }
        cq.prune_n = uint32_t(K);
        for (uint32_t li : leaf_lists) cq.prune_mask |= 1ull << li;
    }

    // Rich simple queries (DSimple2, k_scan_simple<2, true>): <= 4 single-list posting leaves in a tree of depth <= 2, no filter, no facets,
    // sink stages that need membership in <= 4 id lists, the leaves' own presence, or a gather by doc id.
    // DSimple2::ub / prune: per set of present leaves an upper bound of what the score tree and the column boosts can give a doc (see device_types.hpp).
    // Everything is taken on list / column maxima in double precision and handed over 1e-5 above: the kernel's f32 arithmetic on real values stays below.
    void rich_bounds(DSimple2& S, const std::vector<uint16_t>& leaves, uint32_t n) {
        S.prune = 0;
        for (float& u : S.ub) u = std::numeric_limits<float>::infinity();
        static const bool off = std::getenv("VQ_NO_RICH_PRUNE") != nullptr;
        if (off || !cq.facets.empty() || cq.top_k < 1 || col_store.size() != cq.cols.size()) return;
        double v[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < n; ++k) {
            const HList& l = cq.lists[leaves[k]];
            if (l.flags & LIST_F32) v[k] = l.max_value;
            else if (l.max_raw >= 0x7C00 || !(l.term_score >= 0.0f)) return;
            else {
                const uint16_t h = l.max_raw;  // finite non-negative f16
                const int e = (h >> 10) & 31, m = h & 1023;
                v[k] = double(l.term_score) * ((e ? std::ldexp(1.0 + m / 1024.0, e - 15) : std::ldexp(m / 1024.0, -14)) / 100.0);
            }
            if (!(v[k] >= 0.0) || std::isinf(v[k])) return;
        }
        for (uint32_t g = 0; g < S.n_grp; ++g)
            if (!(S.grp_mult[g] >= 0.0f) || std::isinf(S.grp_mult[g])) return;
        for (uint32_t t = 0; t < S.n_tb; ++t)
            if (!(S.tb_mult[t] >= 0.0f) || std::isinf(S.tb_mult[t])) return;
        // the column boosts as monotone maps of the score (boost.rs:283-377): a doc may have no value (the score stays), else the factor / summand is
        // largest at the column's largest value — provided no value turns a factor negative
        struct Col {
            int fun;
            double hi;  // vmax + param
            bool any;
        };
        std::vector<Col> colb;
        for (size_t k = 0; k < cq.cols.size(); ++k) {
            const DColBoost& cb = cq.cols[k];
            const BoostColumn& bc = *col_store[k];
            if (cb.nskip || cb.expr_op != EX_NONE || bc.has_nan) return;
            const double lo = double(bc.vmin) + double(cb.param), hi = double(bc.vmax) + double(cb.param);
            if (bc.any_value) {
                if (cb.fun == BF_MULTIPLY && !(lo >= 0.0)) return;
                if ((cb.fun == BF_LOG10 || cb.fun == BF_LOG2) && !(lo >= 1.0)) return;
                if (std::isinf(hi) || hi != hi) return;
            }
            colb.push_back({cb.fun, hi, bc.any_value});
        }
        for (uint32_t pm = 0; pm < (1u << n); ++pm) {
            double gv[4] = {0, 0, 0, 0};
            bool gp[4] = {false, false, false, false};
            for (uint32_t g = 0; g < S.ngroups; ++g) {
                const uint32_t gm = S.g_mask[g], kind = S.g_kind[g];
                if (kind == OP_AND) {
                    for (uint32_t k = 0; k < n; ++k)
                        if ((gm >> k) & 1u) gv[g] += v[k];
                    gp[g] = (pm & gm) == gm;
                } else if (kind == OP_OR) {
                    double sum = 0.0, nd = 0.0;
                    for (uint32_t sl = 0; sl < S.g_nslots[g]; ++sl) {
                        double m = 0.0;
                        bool any = false;
                        for (uint32_t k = 0; k < n; ++k)
                            if (((gm >> k) & 1u) && S.g_slot[g][k] == sl && ((pm >> k) & 1u)) {
                                m = std::max(m, v[k]);
                                any = true;
                            }
                        if (any) nd += 1.0;  // (at most: a slot counts from 1e-5 on)
                        sum += m;
                    }
                    gv[g] = sum * nd * nd;
                    gp[g] = (pm & gm) != 0;
                } else {  // a leaf, or one leaf over several lists (the largest present value)
                    for (uint32_t k = 0; k < n; ++k)
                        if (((gm >> k) & 1u) && ((pm >> k) & 1u)) gv[g] = std::max(gv[g], v[k]);
                    gp[g] = (pm & gm) != 0;
                }
            }
            double s = 0.0;
            bool present;
            if (S.root_kind == OP_AND) {
                present = true;
                for (uint32_t g = 0; g < S.ngroups; ++g) {
                    present = present && gp[g];
                    s += gv[g];
                }
            } else if (S.root_kind == OP_OR) {
                present = false;
                double sum = 0.0, nd = 0.0;
                for (uint32_t sl = 0; sl < S.root_nslots; ++sl) {
                    double m = 0.0;
                    bool any = false;
                    for (uint32_t g = 0; g < S.ngroups; ++g)
                        if (S.r_slot[g] == sl && gp[g]) {
                            m = std::max(m, gv[g]);
                            any = true;
                        }
                    if (any) nd += 1.0;
                    sum += m;
                    present = present || any;
                }
                s = sum * nd * nd;
            } else {
                present = gp[0];
                s = gv[0];
            }
            if (!present) continue;  // (no hit has this set of leaves: the bound stays +inf)
            for (const Col& c : colb) {
                if (!c.any) continue;
                double b = s;
                switch (c.fun) {
                    case BF_LOG10: b = s * std::log10(c.hi); break;
                    case BF_LOG2: b = s * std::log2(c.hi); break;
                    case BF_MULTIPLY: b = s * c.hi; break;
                    case BF_ADD: b = s + c.hi; break;
                    case BF_REPLACE: b = c.hi; break;
                    default: break;
                }
                s = std::max(s, b);
            }
            const double up = s * (1.0 + 1e-5) + 1e-30;
            float f = up < 3.0e38 ? float(up) : std::numeric_limits<float>::infinity();
            if (double(f) < up) f = std::nextafter(f, std::numeric_limits<float>::infinity());
            S.ub[pm] = f;
        }
        S.prune = 1;
    }

    void detect_rich_simple() {
        static const bool off = std::getenv("VQ_FORCE_GENERIC") != nullptr || std::getenv("VQ_NO_RICH") != nullptr;
        if (off || !count_reqs.empty() || cq.ops.empty() || cq.facets.size() > 2) return;
        // a filter that is one leaf (one term, possibly several id lists: case variants) is a side-list membership test
        if (!cq.fops.empty() && !(cq.fops.size() == 1 && cq.fops[0].kind == OP_LEAF && cq.fops[0].list_count <= 4)) return;
        if (uint64_t(idx.doc_hi) - idx.doc_lo < 65536 || cq.n_top_cols != cq.cols.size() || cq.cols.size() > 4) return;
        DSimple2 S{};
        std::vector<uint16_t> leaves;  // list index of leaf k
        struct Node {
            bool is_leaf;
            uint32_t leaf;      // leaf index
            DOp op;             // group op
            std::vector<uint32_t> kids;  // leaf indices of a group
        };
        std::vector<Node> st;
        bool have_root = false;
        for (size_t o = 0; o < cq.ops.size(); ++o) {
            const DOp& op = cq.ops[o];
            if (op.kind == OP_LEAF) {
                if (op.list_count == 0 || op.list_count > 4) return;
                Node g{false, 0, op, {}};  // a leaf over 2-4 posting lists (a fuzzy term that matched a few dictionary terms): a group that takes the maximum
                g.op.kind = OP_LEAFMAX;
                for (uint32_t j = 0; j < op.list_count; ++j) {
                    const HList& l = cq.lists[op.list_begin + j];
                    if (!(l.flags & LIST_HAS_SCORES) || l.inline_idx >= 0 || l.inline_val_idx >= 0) return;
                    if (leaves.size() >= 4) return;
                    if (l.flags & LIST_F32) S.f32_mask |= uint8_t(1u << leaves.size());
                    g.kids.push_back(uint32_t(leaves.size()));
                    leaves.push_back(uint16_t(op.list_begin + j));
                }
                if (op.list_count == 1) st.push_back(Node{true, g.kids[0], op, {}});
                else st.push_back(g);
            } else if (op.kind == OP_AND || op.kind == OP_OR) {
                if (op.nchild > st.size() || op.nchild > 4) return;
                std::vector<Node> kids(st.end() - op.nchild, st.end());
                st.resize(st.size() - op.nchild);
                bool all_leaves = true;
                for (auto& k : kids) all_leaves = all_leaves && k.is_leaf;
                const bool last = o + 1 == cq.ops.size();
                if (all_leaves && !last) {  // a group of leaves
                    Node g{false, 0, op, {}};
                    for (auto& k : kids) g.kids.push_back(k.leaf);
                    st.push_back(g);
                } else if (last && st.empty()) {  // the root over leaves and groups
                    if (kids.size() > 4) return;
                    S.ngroups = uint8_t(kids.size());
                    S.root_kind = op.kind;
                    S.root_nslots = op.nslots;
                    for (size_t g = 0; g < kids.size(); ++g) {
                        S.r_order[g] = op.and_order[g];
                        S.r_slot[g] = op.child_slot[g];
                        if (kids[g].is_leaf) {
                            S.g_kind[g] = OP_LEAF;
                            S.g_mask[g] = uint8_t(1u << kids[g].leaf);
                        } else {
                            const DOp& gop = kids[g].op;
                            S.g_kind[g] = gop.kind;
                            S.g_nslots[g] = gop.nslots;
                            for (size_t c = 0; c < kids[g].kids.size(); ++c) {
                                const uint32_t leaf = kids[g].kids[c];
                                S.g_mask[g] |= uint8_t(1u << leaf);
                                S.g_order[g][c] = uint8_t(kids[g].kids[gop.and_order[c]]);  // and_order holds child positions
                                S.g_slot[g][leaf] = gop.child_slot[c];
                            }
                        }
                    }
                    have_root = true;
                } else return;  // deeper than two levels
            } else return;  // OP_BOOST1N ...
        }
        if (!have_root) {
            if (st.size() != 1 || !(st[0].is_leaf || st[0].op.kind == OP_LEAFMAX)) return;  // a single leaf
            S.ngroups = 1;
            S.root_kind = OP_LEAF;
            S.root_nslots = 1;
            S.g_kind[0] = st[0].is_leaf ? OP_LEAF : OP_LEAFMAX;
            S.g_mask[0] = 0;
            if (st[0].is_leaf) S.g_mask[0] = 1;
            else
                for (uint32_t leaf : st[0].kids) S.g_mask[0] |= uint8_t(1u << leaf);
        }
        const uint32_t n = uint32_t(leaves.size());
        for (uint32_t k = 0; k < n; ++k) S.leaf_list[k] = leaves[k];
        // side lists of the sink stages
        std::vector<uint16_t> sides;
        auto leaf_of = [&](uint32_t li) -> int {
            for (uint32_t k = 0; k < n; ++k)
                if (leaves[k] == li) return int(k);
            return -1;
        };
        auto side_of = [&](uint32_t li) -> int {
            if (cq.lists[li].flags & LIST_HAS_SCORES) return -1;  // value lists (1:n boosts, precomputed locality) need the interpreter
            for (size_t s2 = 0; s2 < sides.size(); ++s2)
                if (sides[s2] == li) return int(s2);
            if (sides.size() >= 4) return -1;
            sides.push_back(uint16_t(li));
            return int(sides.size() - 1);
        };
        if (cq.groups.size() > 4 || cq.tboosts.size() > 4 || cq.locf.size() > 2) return;
        if (!cq.fops.empty()) {
            S.has_filter = 1;
            for (uint32_t j = 0; j < cq.fops[0].list_count; ++j) {
                const int s2 = side_of(cq.fops[0].list_begin + j);
                if (s2 < 0) return;
                S.filter_mask |= uint8_t(1u << s2);
            }
        }
        for (size_t g = 0; g < cq.groups.size(); ++g) {
            for (uint32_t j = 0; j < cq.groups[g].list_count; ++j) {
                const int s2 = side_of(cq.groups[g].list_begin + j);
                if (s2 < 0) return;
                S.grp_mask[g] |= uint8_t(1u << s2);
            }
            S.grp_mult[g] = cq.groups[g].mult;
        }
        S.n_grp = uint8_t(cq.groups.size());
        for (size_t t = 0; t < cq.tboosts.size(); ++t) {
            const int s2 = side_of(cq.tboosts[t].list);
            if (s2 < 0) return;
            S.tb_side[t] = uint8_t(s2);
            S.tb_mult[t] = cq.tboosts[t].mult;
        }
        S.n_tb = uint8_t(cq.tboosts.size());
        for (size_t f = 0; f < cq.locf.size(); ++f) {
            if (cq.locf[f].list_count == kLocPrecomputed) return;
            for (uint32_t j = 0; j < cq.locf[f].list_count; ++j) {
                const uint32_t li = cq.loc_idx[cq.locf[f].list_begin + j];
                const int k = leaf_of(li);
                if (k >= 0) {
                    if (S.loc_leaf[f] & (1u << k)) return;  // the same list twice: counted twice by the reference, not representable as a mask
                    S.loc_leaf[f] |= uint8_t(1u << k);
                } else {
                    const int s2 = side_of(li);
                    if (s2 < 0 || (S.loc_side[f] & (1u << s2))) return;
                    S.loc_side[f] |= uint8_t(1u << s2);
                }
            }
        }
        S.n_loc = uint8_t(cq.locf.size());
        S.n_side = uint8_t(sides.size());
        for (size_t s2 = 0; s2 < sides.size(); ++s2) S.side_list[s2] = sides[s2];
        for (uint32_t li = 0; li < cq.lists.size(); ++li)  // every list must be a leaf or a side list
            if (leaf_of(li) < 0 && std::find(sides.begin(), sides.end(), uint16_t(li)) == sides.end()) return;
        // flags as for the flat simple queries, per leaf
        uint32_t f = (1u << 17) | (1u << 18);
        bool seq = false;
        for (uint32_t k = 0; k < n; ++k) {
            const HList& l = cq.lists[leaves[k]];
            if ((l.flags & LIST_COVER) && (l.flags & LIST_BITMAP)) seq = true;
        }
        if (seq) f |= 1u << 16;
        bool any_cover = false;
        for (uint32_t k = 0; k < n; ++k) {
            const HList& l = cq.lists[leaves[k]];
            const bool cover = l.flags & LIST_COVER;
            any_cover = any_cover || cover;
            if (cover) f |= 1u << (8 + k);
            if ((l.flags & LIST_BITMAP) && (seq || !cover)) f |= 1u << k;
            if (uint64_t(l.len) * 8192 >= 200 * (uint64_t(idx.doc_hi) - idx.doc_lo)) f |= 1u << (20 + k);
        }
        if (!any_cover) return;
        rich_bounds(S, leaves, n);
        cq.simple2 = S;
        cq.simple_n = n;
        cq.simple_flags = f;
    }

    // Wide queries (DWide, k_scan_wide): 5..16 single-list posting leaves in a tree of depth <= 2, nothing but the score tree — the
    // query generator's shapes (one leaf per term and field).
    void detect_wide() {
        static const bool off = std::getenv("VQ_FORCE_GENERIC") != nullptr || std::getenv("VQ_NO_WIDE") != nullptr;
        if (off || !count_reqs.empty() || cq.ops.empty() || !cq.fops.empty() || !cq.groups.empty() || !cq.tboosts.empty() || !cq.cols.empty() || !cq.locf.empty() ||
            !cq.facets.empty() || uint64_t(idx.doc_hi) - idx.doc_lo < 65536)
            return;
        DWide W{};
        struct Node {
            bool is_leaf;
            uint32_t leaf;
            DOp op;
            std::vector<uint32_t> kids;  // leaf indices of a group, in child order
        };
        std::vector<Node> st;
        std::vector<uint16_t> leaves;
        bool have_root = false;
        std::vector<Node> root_kids;
        DOp root_op{};
        for (size_t o = 0; o < cq.ops.size(); ++o) {
            const DOp& op = cq.ops[o];
            if (op.kind == OP_LEAF) {
                if (op.list_count != 1) return;
                const HList& l = cq.lists[op.list_begin];
                if (!(l.flags & LIST_HAS_SCORES) || l.inline_idx >= 0 || l.inline_val_idx >= 0) return;
                if (leaves.size() >= size_t(kWideMax)) return;
                for (uint16_t e : leaves)
                    if (e == op.list_begin) return;  // one list under two leaves: the count planes would count it once
                st.push_back(Node{true, uint32_t(leaves.size()), op, {}});
                leaves.push_back(op.list_begin);
            } else if (op.kind == OP_AND || op.kind == OP_OR) {
                if (op.nchild > st.size() || op.nchild > kWideMax) return;
                std::vector<Node> kids(st.end() - op.nchild, st.end());
                st.resize(st.size() - op.nchild);
                bool all_leaves = true;
                for (auto& k : kids) all_leaves = all_leaves && k.is_leaf;
                const bool last = o + 1 == cq.ops.size();
                if (last && st.empty()) {
                    root_kids = kids;
                    root_op = op;
                    have_root = true;
                } else if (all_leaves) {
                    Node g{false, 0, op, {}};
                    for (auto& k : kids) g.kids.push_back(k.leaf);
                    for (size_t c = 1; c < g.kids.size(); ++c)
                        if (g.kids[c] != g.kids[c - 1] + 1) return;  // (postfix order: always consecutive)
                    st.push_back(g);
                } else return;  // deeper than two levels
            } else return;  // OP_BOOST1N
        }
        if (!have_root || leaves.size() < 5 || root_kids.size() > size_t(kWideMax)) return;  // (<= 4 leaves: the simple kernels)
        const uint32_t n = uint32_t(leaves.size());
        W.n_leaves = uint8_t(n);
        W.n_groups = uint8_t(root_kids.size());
        W.root_kind = root_op.kind;
        W.root_nslots = root_op.nslots;
        for (uint32_t k = 0; k < n; ++k) W.leaf_list[k] = leaves[k];
        for (size_t g = 0; g < root_kids.size(); ++g) {
            const Node& kid = root_kids[g];
            W.r_and_order[g] = root_op.and_order[g];
            W.r_slot[g] = root_op.child_slot[g];
            if (kid.is_leaf) {
                W.g_kind[g] = OP_LEAF;
                W.g_begin[g] = uint8_t(kid.leaf);
                W.g_count[g] = 1;
                W.leaf_and_order[kid.leaf] = uint8_t(kid.leaf);
                W.leaf_slot_order[kid.leaf] = uint8_t(kid.leaf);
            } else {
                W.g_kind[g] = kid.op.kind;
                W.g_begin[g] = uint8_t(kid.kids[0]);
                W.g_count[g] = uint8_t(kid.kids.size());
                std::vector<uint32_t> by_slot;
                for (size_t c = 0; c < kid.kids.size(); ++c) {
                    W.leaf_slot[kid.kids[c]] = kid.op.child_slot[c];
                    W.leaf_and_order[kid.kids[0] + c] = uint8_t(kid.kids[kid.op.and_order[c]]);  // and_order holds child positions
                    by_slot.push_back(kid.kids[c]);
                }
                std::stable_sort(by_slot.begin(), by_slot.end(), [&](uint32_t a, uint32_t b) { return W.leaf_slot[a] < W.leaf_slot[b]; });
                for (size_t c = 0; c < by_slot.size(); ++c) W.leaf_slot_order[kid.kids[0] + c] = uint8_t(by_slot[c]);
            }
        }
        {
            std::vector<uint32_t> by_slot;
            for (uint32_t g = 0; g < W.n_groups; ++g) by_slot.push_back(g);
            std::stable_sort(by_slot.begin(), by_slot.end(), [&](uint32_t a, uint32_t b) { return W.r_slot[a] < W.r_slot[b]; });
            for (uint32_t g = 0; g < W.n_groups; ++g) W.r_slot_order[g] = uint8_t(by_slot[g]);
        }
        for (uint32_t li = 0; li < cq.lists.size(); ++li)  // every list must be a leaf
            if (std::find(leaves.begin(), leaves.end(), uint16_t(li)) == leaves.end()) return;
        bool seq = false, any_cover = false;
        for (uint32_t k = 0; k < n; ++k) {
            const HList& l = cq.lists[leaves[k]];
            if ((l.flags & LIST_COVER) && (l.flags & LIST_BITMAP)) seq = true;
        }
        for (uint32_t k = 0; k < n; ++k) {
            const HList& l = cq.lists[leaves[k]];
            const bool cover = l.flags & LIST_COVER;
            any_cover = any_cover || cover;
            if (cover) W.cover_mask |= uint16_t(1u << k);
            if ((l.flags & LIST_BITMAP) && (seq || !cover)) W.bitmap_mask |= uint16_t(1u << k);
            if (l.flags & LIST_F32) W.f32_mask |= uint16_t(1u << k);
            if (uint64_t(l.len) * 8192 >= 200 * (uint64_t(idx.doc_hi) - idx.doc_lo)) W.prefetch_mask |= uint16_t(1u << k);
        }
        if (!any_cover) return;
        W.seq = seq ? 1 : 0;
        cq.wide = W;
        cq.simple_flags = 1u << 24;
        cq.simple_n = 0;
    }

    // ------------------------------------------------------------ the whole request (search.rs:143-228)
    void run() {
        cq.lists.reserve(8);  // (the vectors of a typical request: one allocation each instead of a growth chain)
        cq.ops.reserve(8);
        cq.pres.reserve(8);
        cq.pres_in.reserve(16);
        by_address.reserve(8);
        // (`select` itself is not looked at: search::search leaves the reading of the selected fields to to_documents, search.rs:63-103; together with
        //  why_found it asks for why_found_info, :220-224 — complete_why_found_requests)
        // (`suggest` is not looked at either: search::search reads search_req, search.rs:151-155; suggest_multi is vq_suggest_json)
        if (!req.search_req) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"search_req is None, but is required in search\" ");
        const uint64_t top64 = req.top.value_or(10), skip64 = req.skip.value_or(0);  // :146
        const uint64_t want = top64 + skip64 < top64 ? ~0ull : top64 + skip64;
        cq.top = uint32_t(std::min<uint64_t>(top64, kMaxTopK));
        cq.skip = uint32_t(std::min<uint64_t>(skip64, kMaxTopK));
        cq.key_upper = req.key_upper;
        if (want > uint64_t(kMaxTopK)) {  // deep request: rank the best kMaxTopK here, the caller pages on below the last key
            cq.deep = true;
            cq.top = uint32_t(kMaxTopK);
            cq.skip = 0;
        }
        cq.top_k = uint32_t(std::max<uint64_t>(uint64_t(cq.top) + cq.skip, 1));

        // collect_all_field_request_into_cache (execution_plan.rs:91-106), then the flags set during plan creation
        if (req.phrase_boosts)
            for (auto& el : *req.phrase_boosts) {
                add_to_cache(el.search1, false);
                add_to_cache(el.search2, false);
            }
        collect(*req.search_req, false);
        if (req.filter) collect(*req.filter, true);
        if (req.filter) flag_tree(*req.filter);
        flag_tree(*req.search_req);
        if (req.why_found) flag_tree_texts(*req.search_req);
        if (req.phrase_boosts)
            for (auto& el : *req.phrase_boosts) {
                leaf(el.search1).get_ids = true;
                leaf(el.search2).get_ids = true;
            }

        {  // explain: the reference's set operations look at the flag of ONE operand's request (set_op.rs:120,384); supported when
           // all leaves agree — request.explain, or a single leaf with options.explain (tests/all/tests.rs:348-364)
            size_t leaves = 0, flagged = 0;
            std::function<void(const SearchRequest&)> walk = [&](const SearchRequest& r) {
                if (r.kind == SearchRequest::Search) {
                    ++leaves;
                    flagged += part_explains(r.part) ? 1 : 0;
                } else
                    for (auto& q : r.tree.queries) walk(q);
            };
            walk(*req.search_req);
            if (flagged && flagged != leaves) unsupported("explain on a part of the query tree");
            explain_on = flagged != 0;
            if (explain_on && req.phrase_boosts)  // apply_boost_from_iter records a boost or not depending on where its look-ahead rests (boost.rs:213-228): on ALL hits
                unsupported("explain with phrase_boosts");
            if (explain_on) xplan = std::make_shared<ExplainPlan>();
        }

        // filter tree (ids only), then the score tree
        uint32_t sp = 0;
        if (req.filter) {
            compile_node(*req.filter, true, cq.fops, sp, {});
            if (cq.fops.empty()) unsupported("filter that reduces to nothing");
        }
        sp = 0;
        NodeInfo root = compile_node(*req.search_req, false, cq.ops, sp, req.boost.value_or(std::vector<RequestBoostPart>{}));
        for (uint32_t li : root.cover) cq.lists[li].flags |= LIST_COVER;

        // anchor-level column boosts (execution_plan.rs:175-189, boost.rs:470-504)
        if (req.boost)
            for (auto& b : *req.boost) {
                if (b.path.find("[]") != std::string::npos) continue;  // only used through a matching 1:n search path
                auto it = idx.boost.find(b.path + BOOST_VALID_TO_VALUE);
                if (it == idx.boost.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + b.path + BOOST_VALID_TO_VALUE);
                DColBoost cb{};
                cb.values = it->second.values.as<float>();
                cb.present = it->second.has_present ? it->second.present.as<uint32_t>() : nullptr;
                cb.key_base = it->second.key_base;
                cb.num_keys = it->second.num_keys;
                fill_boost_params(cb, b);
                cq.cols.push_back(cb);
                col_store.push_back(&it->second);
                cq.algorithmic_bytes += 0;  // 4 B gather per hit, unknown until run time
            }
        cq.n_top_cols = uint32_t(cq.cols.size());
        if (explain_on) {
            xplan->root = root.ex_nodes.at(0);
            xplan->cols = cq.cols;  // (the request-level boosts only: 1:n boosts are declined above)
            uint32_t depth = 0, deepest = 0;
            ex_emit(xplan->root, depth, deepest);
            if (deepest > kExStack) unsupported("explain: more than " + std::to_string(kExStack) + " operands alive at once");
            cq.explain_plan = xplan;
        }
        if (!cq.leaf_cols.empty()) {  // parameters of the OP_BOOST1N ops live behind the request-level boosts
            if (cq.cols.size() + cq.leaf_cols.size() > 255) unsupported("more than 255 boosts in one query");
            for (DOp& op : cq.ops)
                if (op.kind == OP_BOOST1N) op.child_slot[0] = uint8_t(op.child_slot[0] + cq.n_top_cols);
            cq.cols.insert(cq.cols.end(), cq.leaf_cols.begin(), cq.leaf_cols.end());
        }

        // phrase boosts (execution_plan.rs:202-262, plan_steps.rs:235-293, search_field.rs:247-275)
        if (req.phrase_boosts) {
            std::map<std::pair<std::string, std::string>, std::vector<uint32_t>> grouped;  // (term1, term2) -> lists
            for (auto& pb : *req.phrase_boosts) {
                Leaf& l1 = field_result(pb.search1);
                Leaf& l2 = field_result(pb.search2);
                if (pb.search1.path != pb.search2.path) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"phrase boost over two different paths\" ");
                std::string path = pb.search1.path;
                if (!ends_with(path, TEXTINDEX)) path += TEXTINDEX;
                path += PHRASE_PAIR_TO_ANCHOR;
                auto it = idx.phrase.find(path);
                if (it == idx.phrase.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
                const PhraseStore& store = it->second;
                auto& lists = grouped[{pb.search1.terms[0], pb.search2.terms[0]}];
                for (uint32_t t1 : l1.hits_ids)
                    for (uint32_t t2 : l2.hits_ids) {
                        auto key = std::make_pair(t1, t2);
                        auto kit = std::lower_bound(store.keys.begin(), store.keys.end(), key);
                        if (kit == store.keys.end() || *kit != key) continue;
                        const size_t k = size_t(kit - store.keys.begin());
                        HList h;
                        h.d_docs = store.anchors.as<uint32_t>() + store.start[k];
                        h.len = store.len[k];
                        h.global_len = store.len[k];
                        lists.push_back(add_list(h));
                        cq.algorithmic_bytes += 4ull * h.len;
                    }
            }
            // lists of one group must be contiguous for DGroup: re-emit them in group order
            for (auto& [terms, lists] : grouped) {
                if (lists.empty()) continue;
                bool contiguous = true;
                for (size_t i = 1; i < lists.size(); ++i) contiguous = contiguous && lists[i] == lists[i - 1] + 1;
                uint32_t begin = lists[0];
                if (!contiguous) {
                    begin = uint32_t(cq.lists.size());
                    for (uint32_t li : lists) {
                        HList copy = cq.lists[li];
                        cq.total_len -= copy.len;  // counted twice otherwise
                        add_list(copy);
                    }
                    for (uint32_t li : lists) cq.lists[li].len = 0, cq.lists[li].d_docs = nullptr;
                }
                DGroup g{};
                g.list_begin = uint16_t(begin);
                g.list_count = uint16_t(lists.size());
                g.mult = 5.0f;  // plan_steps.rs:270-272
                cq.groups.push_back(g);
            }
        }

        // boost_term (search.rs:176-178, boost.rs:89-195, 380-402)
        if (req.boost_term)
            for (auto& part : *req.boost_term) {
                Leaf l;
                l.part = &part;
                lookup_terms(idx, l, false, true);
                const float mult = part.boost.value_or(2.0f);  // boost.rs:393
                for (uint32_t li : ids_to_anchor_lists(l, true)) {
                    DTermBoost tb{};
                    tb.list = uint16_t(li);
                    tb.mult = mult;
                    cq.tboosts.push_back(tb);
                    cq.algorithmic_bytes += 4ull * cq.lists[li].len;
                }
            }

        if (req.why_found && req.has_select) {  // search.rs:220-224, why_found.rs:20-31
            if (idx.sharded()) unsupported("why_found with select on a shard (why_found_info joins the returned anchors to their texts: the caller's merged page may hold other shards' anchors)");
            auto plan = std::make_shared<WhyFoundPlan>();
            for (auto& [path, terms] : term_id_hits) {
                std::vector<uint32_t>& all = (*plan)[path];
                for (auto& [term, ids] : terms) all.insert(all.end(), ids.begin(), ids.end());
                std::sort(all.begin(), all.end());
                all.erase(std::unique(all.begin(), all.end()), all.end());
            }
            cq.why_found_plan = std::move(plan);
        }
        // text locality (search.rs:180-184, boost.rs:11-87)
        if (req.text_locality)
            for (auto& [path, terms] : term_id_hits) {
                if (terms.size() <= 1) continue;  // boost.rs:36-39
                const KVStore& t2t = kv_store(path + TOKENS_TO_TEXT_ID);
                if (!idx.is_anchor_identity(path)) {
                    // Text ids are not anchors (boost.rs:72-83): the K7 pre-pass (run_locality_jobs) gathers the terms' token -> text rows, counts
                    // the texts, expands them to anchors and hands back one (anchor, smallest 2*c*c) list for the field
                    const KVStore& t2a = kv_store(path + TEXT_ID_TO_ANCHOR);
                    if (!t2t.text_csr || !t2a.d_row_len.p) unsupported("text_locality: " + path + " is not staged for the device pre-pass");
                    LocalityJob job;
                    job.t2t_path = path + TOKENS_TO_TEXT_ID;
                    job.t2a_path = path + TEXT_ID_TO_ANCHOR;
                    job.key = "loc|" + path;
                    for (auto& [term, ids] : terms) {
                        job.key += "|";
                        for (uint32_t id : ids) {
                            job.tokens.push_back(id);
                            job.key += std::to_string(id) + ",";
                        }
                    }
                    const LocalityJob* done = nullptr;
                    if (localities) {
                        auto it = localities->find(job.key);
                        if (it != localities->end()) done = &it->second;
                    }
                    if (!done) {
                        cq.locality_requests.push_back(std::move(job));  // compiled again after the job ran
                        continue;
                    }
                    if (!done->len) continue;  // no text holds two of the terms' tokens (in this shard)
                    HList h;
                    h.d_docs = done->d_docs;
                    h.d_scores = reinterpret_cast<const uint16_t*>(done->d_vals);
                    h.len = done->len;
                    h.global_len = done->len;
                    h.flags = LIST_HAS_SCORES | LIST_F32;
                    h.term_score = 1.0f;
                    DLocField lf{};
                    lf.list_begin = uint16_t(add_list(h));
                    lf.list_count = kLocPrecomputed;
                    cq.locf.push_back(lf);
                    cq.algorithmic_bytes += 8ull * h.len;
                    continue;
                }
                DLocField lf{};
                lf.list_begin = uint16_t(cq.loc_idx.size());
                uint32_t count = 0;
                const PostingStore* same_ps = nullptr;  // identity column: token->text row t == docs of posting list t (checked when staged)
                if (t2t.rows_equal_postings) {
                    auto pit = idx.postings.find(path + TO_ANCHOR_ID_SCORE);
                    if (pit != idx.postings.end()) same_ps = &pit->second;
                }
                for (auto& [term, ids] : terms)
                    for (uint32_t id : ids) {
                        if (id < t2t.key_base || id - t2t.key_base >= t2t.num_keys) continue;
                        const uint32_t r = id - t2t.key_base;
                        if (t2t.host_off[r] == t2t.host_off[r + 1]) continue;
                        HList h;
                        if (same_ps && id < same_ps->num_tokens) {  // read the posting list's doc ids (and its bitmap image) instead of a second copy
                            h.d_docs = same_ps->docs.as<uint32_t>() + same_ps->start[id];
                            h.len = same_ps->len[id];
                            if (!same_ps->bm_start.empty() && same_ps->bm_start[id] >= 0) {
                                h.flags |= LIST_BITMAP;
                                h.d_bitmap = same_ps->bitmaps.as<uint32_t>() + same_ps->bm_start[id];
                                h.d_rank_dir = same_ps->rank_dir.as<uint32_t>() + same_ps->rd_start[id];
                            }
                        } else {
                            h.d_docs = t2t.values.as<uint32_t>() + t2t.start[r];
                            h.len = t2t.len[r];
                        }
                        h.global_len = t2t.host_off[r + 1] - t2t.host_off[r];
                        uint32_t li = UINT32_MAX;
                        for (uint32_t e = 0; e < cq.lists.size(); ++e)  // already a list of this query (the term's own posting leaf): share its tile bitmap
                            if (cq.lists[e].d_docs == h.d_docs && cq.lists[e].inline_idx < 0 && cq.lists[e].len == h.len) {
                                li = e;
                                break;
                            }
                        if (li == UINT32_MAX) {
                            li = add_list(h);
                            cq.algorithmic_bytes += 4ull * h.len;
                        }
                        cq.loc_idx.push_back(uint16_t(li));
                        ++count;
                    }
                lf.list_count = uint16_t(count);
                cq.locf.push_back(lf);
            }

        // facets (search.rs:188-206, facet.rs:31-73)
        if (req.facets)
            for (auto& fr : *req.facets) {
                std::vector<std::string> steps = get_steps_to_anchor(fr.field);
                std::string store_path;
                if (steps.size() == 1) store_path = steps.front() + PARENT_TO_VALUE_ID;
                else if (idx.kv.count(steps.back() + ANCHOR_TO_TEXT_ID)) store_path = steps.back() + ANCHOR_TO_TEXT_ID;
                const KVStore& kv = store_path.empty() ? idx.composed_facet(steps) : kv_store(store_path);  // facet.rs:38-57 / :59-70
                if (!kv.facet_csr) unsupported("facet source " + store_path + " is not staged as an anchor-keyed CSR");
                auto dit = idx.dict.find(steps.back());
                if (dit == idx.dict.end()) throw VelociError(ERR_FST_NOT_FOUND, "fst not found loaded in indices " + steps.back() + " ");
                DFacet f{};
                f.offsets = kv.csr_off.as<uint64_t>();
                f.values = kv.csr_values.as<uint32_t>();
                f.direct = kv.csr_direct.p ? kv.csr_direct.as<uint32_t>() : nullptr;
                f.key_base = kv.csr_key_base;
                f.num_keys = kv.csr_num_keys;
                // value ids beyond the dictionary (texts longer than do_not_store_text_longer_than) are counted too and render as ""
                f.num_values = std::max<uint32_t>(uint32_t(dit->second.terms.size()), kv.csr_num_keys ? kv.csr_max_value + 1 : 0);
                // top: null reports every value that was counted (facet.rs:19-23 truncates only with Some(top))
                const uint64_t want = fr.top ? std::min<uint64_t>(*fr.top, f.num_values) : f.num_values;
                // k_facet_select ranks up to kMaxTopK entries per facet; beyond that the histogram itself goes to the host (finish_batch)
                f.top = want > uint64_t(kMaxTopK) ? 0u : uint32_t(want);
                cq.facets.push_back(f);
                FacetOut fo;
                fo.field = fr.field;
                fo.dict_path = steps.back();
                fo.top = f.top;
                fo.host_top = want > uint64_t(kMaxTopK) ? uint32_t(want) : 0u;
                fo.num_values = f.num_values;
                cq.facet_out.push_back(fo);
                cq.algorithmic_bytes += 4ull * f.num_values;
            }
        cq.algorithmic_bytes += 8ull * cq.top_k;

        if (label_wanted) {  // some OR needs the label of a nested AND: that takes the sizes of the two-operand ANDs as well
            if (maybe_reqs.empty() && count_reqs.empty()) unsupported("OR over an operand whose term label is only known at run time (set_op.rs:143,439)");
            if (!idx.can_sum_over_shards())
                unsupported("OR over an AND whose label follows run-time result sizes, on a sharded index without vq_index_set_allreduce");
            count_reqs.insert(count_reqs.end(), maybe_reqs.begin(), maybe_reqs.end());
        }
        if (2 * count_reqs.size() + 1 > 256) unsupported("more than 127 AND operands whose sizes must be measured first");
        build_presence_program();
        if (!count_reqs.empty())  // count pre-pass: every tile that holds a doc of any list is visited
            for (auto& l : cq.lists) l.flags |= LIST_COVER;
        {  // shape that the kernel scores without the interpreter: <= 4 single-list posting leaves under one AND/OR
            const size_t n = cq.ops.size();
            auto is_leaf1 = [&](const DOp& o) {
                return o.kind == OP_LEAF && o.list_count == 1 && (cq.lists[o.list_begin].flags & LIST_HAS_SCORES) && !(cq.lists[o.list_begin].flags & LIST_F32);
            };
            if (n == 1 && is_leaf1(cq.ops[0])) cq.simple_n = 1;
            else if (n >= 3 && n <= 5 && cq.ops[n - 1].kind != OP_LEAF && cq.ops[n - 1].nchild == n - 1) {
                bool ok = true;
                for (size_t i = 0; i + 1 < n; ++i) ok = ok && is_leaf1(cq.ops[i]);
                if (ok) cq.simple_n = uint32_t(n - 1);
            }
            // OR whose operands all have their own term slot: order the leaves by slot, so that the kernels' slot loop
            // (set_op.rs:169-186) is the plain left-to-right sum over the operands
            if (cq.simple_n > 1 && cq.ops[n - 1].kind == OP_OR && cq.ops[n - 1].nslots == cq.simple_n) {
                DOp& root = cq.ops[n - 1];
                std::vector<std::pair<uint8_t, DOp>> leaves;
                for (uint32_t k = 0; k < cq.simple_n; ++k) leaves.push_back({root.child_slot[k], cq.ops[k]});
                std::stable_sort(leaves.begin(), leaves.end(), [](auto& x, auto& y) { return x.first < y.first; });
                for (uint32_t k = 0; k < cq.simple_n; ++k) {
                    cq.ops[k] = leaves[k].second;
                    root.child_slot[k] = leaves[k].first;
                }
            }
        }

        {  // pure simple queries run on k_scan_simple (fixed 8192-doc tiles)
            static const bool force_generic = std::getenv("VQ_FORCE_GENERIC") != nullptr;
            const bool pure = cq.simple_n && count_reqs.empty() && cq.fops.empty() && cq.groups.empty() && cq.tboosts.empty() && cq.cols.empty() && cq.locf.empty() &&
                              cq.facets.empty() && uint64_t(idx.doc_hi) - idx.doc_lo >= 65536;
            if (pure && !force_generic) {
                uint32_t f = 1u << 17;
                // An AND whose cover (sparsest operand) is ONE list with a tile directory (at least 1/4096 of the docs) and whose other operands
                // all have bitmap images: k_scan_probe streams the cover's postings — as ids and scores, even when the cover has a bitmap
                // image of its own — and tests their bits in the operands' LDS tiles (VQ_NO_PROBE=1: k_scan_simple instead).
                // On shards below ~40 M docs k_scan_simple is still ahead (a span is then a few dozen tiles: the probe kernel's start-up, threshold
                // warm-up and pipeline drain weigh more — 12.5 M docs: 1.13 against 1.55 ms per 1024 queries; 25 M: 2.18 / 2.47; 50 M: 4.34 / 4.19;
                // 100 M: 8.5 / 7.2).  VQ_PROBE_MIN_DOCS moves the line (the tests put it at 0).
                static const bool no_probe = std::getenv("VQ_NO_PROBE") != nullptr;
                static const uint64_t probe_min_docs = std::getenv("VQ_PROBE_MIN_DOCS") ? uint64_t(std::atoll(std::getenv("VQ_PROBE_MIN_DOCS"))) : 40'000'000ull;
                bool probe = !no_probe && cq.simple_n >= 2 && cq.ops[cq.simple_n].kind == OP_AND && uint64_t(idx.doc_hi) - idx.doc_lo >= probe_min_docs;
                uint32_t arr_mask = 0;  // operands probed as 16-bit arrays (2 B per posting) instead of bitmap words (a bit per doc): the fewer bytes win
                if (probe) {
                    static const bool no_arr = std::getenv("VQ_PROBE_NO_ARR") != nullptr;
                    uint32_t covers = 0;
                    for (uint32_t k = 0; k < cq.simple_n; ++k) {
                        const HList& l = cq.lists[cq.ops[k].list_begin];
                        if (l.flags & LIST_COVER) {
                            ++covers;
                            probe = probe && l.d_cov32;
                        } else {
                            const bool arr = l.d_arr16 && !no_arr && (!(l.flags & LIST_BITMAP) || uint64_t(l.len) * 16 < uint64_t(idx.doc_hi) - idx.doc_lo);
                            if (arr) arr_mask |= 1u << k;
                            probe = probe && (arr || (l.flags & LIST_BITMAP));
                        }
                    }
                    probe = probe && covers == 1;
                }
                // An OR of 2 or 3 leaves with a term slot each whose sparsest operand has a tile-packed image and whose other operands are bitmap words, on an
                // unsharded index: k_scan_probe_or counts the union from the words' set bits and scores only the docs that hold the cover (the
                // sparsest operand) — a doc without it scores at most `or_skip_bound`, and finish_batch confirms that the request's k-th best key lies
                // above that before the result is handed out (otherwise the request runs again with exact_routes_only).  Sharded: a rank's partial
                // leaves for the exchange before the host sees it, so shards keep k_scan_simple.  VQ_NO_PROBE_OR=1 switches the route off.
                static const bool no_probe_or = std::getenv("VQ_NO_PROBE_OR") != nullptr;
                uint32_t or_cover = UINT32_MAX;
                if (!probe && !no_probe && !no_probe_or && !req.exact_routes_only && cq.top_k >= 1 && cq.simple_n >= 2 && cq.simple_n <= 3 && cq.ops[cq.simple_n].kind == OP_OR &&
                    cq.ops[cq.simple_n].nslots == cq.simple_n && !idx.sharded() && !(idx.comm && idx.comm->nranks > 1) && uint64_t(idx.doc_hi) - idx.doc_lo >= probe_min_docs) {
                    for (uint32_t k = 0; k < cq.simple_n; ++k) {
                        const HList& l = cq.lists[cq.ops[k].list_begin];
                        if (l.d_cov32 && (or_cover == UINT32_MAX || l.len < cq.lists[cq.ops[or_cover].list_begin].len)) or_cover = k;
                    }
                    double sum = 0.0, cnt = 0.0;
                    for (uint32_t k = 0; k < cq.simple_n && or_cover != UINT32_MAX; ++k) {
                        if (k == or_cover) continue;
                        const HList& l = cq.lists[cq.ops[k].list_begin];
                        if (!(l.flags & LIST_BITMAP) || (l.flags & LIST_F32) || l.max_raw >= 0x7C00 || !(l.term_score >= 0.0f)) {
                            or_cover = UINT32_MAX;
                            break;
                        }
                        const uint16_t h = l.max_raw;  // finite non-negative f16
                        const int e = (h >> 10) & 31, m = h & 1023;
                        const double a = e ? std::ldexp(1.0 + m / 1024.0, e - 15) : std::ldexp(m / 1024.0, -14);
                        const double v = double(l.term_score) * (a / 100.0);
                        if (v >= 0.00001 * (1.0 - 1e-5)) cnt += 1.0;  // (set_op.rs:180: a slot counts from 1e-5 on)
                        sum += v;
                    }
                    if (or_cover != UINT32_MAX) {
                        const HList& c = cq.lists[cq.ops[or_cover].list_begin];
                        if (!(c.term_score > 0.0f) || c.max_raw >= 0x7C00) or_cover = UINT32_MAX;  // (the kernel's bound is taken over the cover's raw scores)
                        else {
                            const double b = sum * cnt * cnt * (1.0 + 1e-5);  // f32 rounding of the kernel's sums stays below the margin
                            cq.or_skip_bound = b < 3.0e38 ? float(b) : std::numeric_limits<float>::infinity();
                            if (cq.or_skip_bound < float(b)) cq.or_skip_bound = std::nextafter(cq.or_skip_bound, std::numeric_limits<float>::infinity());
                            probe = true;
                        }
                    }
                }
                const bool or_probe = or_cover != UINT32_MAX && probe;
                bool seq = false;
                for (uint32_t k = 0; k < cq.simple_n && !probe; ++k) {
                    const HList& l = cq.lists[cq.ops[k].list_begin];
                    if ((l.flags & LIST_COVER) && (l.flags & LIST_BITMAP)) seq = true;
                }
                if (seq) f |= 1u << 16;
                if (probe) {
                    f |= 1u << 25 | arr_mask << 12 | (or_probe ? 1u << 27 : 0u);
                    cq.probe = DProbe{};
                    for (uint32_t k = 0; k < cq.simple_n; ++k) {
                        const HList& l = cq.lists[cq.ops[k].list_begin];
                        cq.probe.leaf[k] = DProbeLeaf{l.d_cov32, l.d_arr16, l.d_gdir};
                    }
                }
                // ... and with top + skip <= 32 (the candidate buffer is one key per lane, the query has a shared pool) the persistent form of that
                // kernel, k_scan_ring: loader waves stream the tiles into LDS rings, consumer waves probe (opt-in)
                // Measured on launches that read no list twice (256 distinct queries per launch, 100 M docs): 1.95 ms against k_scan_probe's 1.94 —
                // the stream side reaches 5.9 TB/s alone, the consumer waves do not keep up: opt-in, VQ_RING=1
                static const bool ring = std::getenv("VQ_RING") != nullptr && std::atoi(std::getenv("VQ_RING")) != 0;
                if (probe && ring && !arr_mask && !or_probe && cq.top_k >= 1 && cq.top_k <= kPoolMaxK) f |= 1u << 26;
                for (uint32_t k = 0; k < cq.simple_n; ++k) {
                    const HList& l = cq.lists[cq.ops[k].list_begin];
                    const bool cover = or_probe ? k == or_cover : (l.flags & LIST_COVER) != 0;
                    if (cover) f |= 1u << (8 + k);
                    if ((l.flags & LIST_BITMAP) && (seq || !cover) && !((arr_mask >> k) & 1u)) f |= 1u << k;
                    if (uint64_t(l.len) * 8192 >= 200 * (uint64_t(idx.doc_hi) - idx.doc_lo)) f |= 1u << (20 + k);
                }
                // a single leaf whose list has a tile-packed image: k_scan_union streams that (4 B per posting) instead of ids + scores (6 B)
                static const bool no_union_cov = std::getenv("VQ_NO_UNION_COV") != nullptr;
                if (cq.simple_n == 1 && !no_union_cov) {
                    const HList& l = cq.lists[cq.ops[0].list_begin];
                    if (l.d_cov32 && l.d_gdir && l.d_tile_dir && l.term_score > 0.0f && !(l.flags & LIST_F32) && l.inline_idx < 0) {
                        f |= 1u << 28;
                        cq.probe = DProbe{};
                        cq.probe.leaf[0] = DProbeLeaf{l.d_cov32, l.d_arr16, l.d_gdir};
                    }
                }
                cq.simple_flags = f;
            }
        }

        {  // one materialised leaf, nothing else that shapes presence or score: k_scan_leaf_f32 streams the list (facets counted per entry)
            static const bool off = std::getenv("VQ_FORCE_GENERIC") != nullptr || std::getenv("VQ_NO_LEAF_F32") != nullptr;
            if (!off && !cq.simple_flags && count_reqs.empty() && cq.ops.size() == 1 && cq.ops[0].kind == OP_LEAF && cq.ops[0].list_count == 1 &&
                (cq.lists[cq.ops[0].list_begin].flags & LIST_F32) && cq.fops.empty() && cq.groups.empty() && cq.tboosts.empty() && cq.cols.empty() && cq.locf.empty() &&
                cq.facets.empty())  // (with facets the rich kernel is faster: its 64-hit rounds put less pressure on the histogram's hot counters
                                    //  than every span adding at once — an LDS-privatised histogram would lift that)
                cq.simple_flags = 1u << 19;
        }
        if (!cq.simple_flags) detect_rich_simple();
        if (!cq.simple_flags) detect_wide();
        if (!cq.simple_flags || ((cq.simple_flags >> 24) & 1u)) compute_prune_table();
        if (!cq.simple_flags && count_reqs.empty()) {  // k_tile_scan: a dense cover list means every tile gets visited anyway: walk them in order
            bool dense_cover = false;                   // and read the dense lists as bitmap images instead of scattering them
            for (auto& l : cq.lists) dense_cover = dense_cover || ((l.flags & LIST_COVER) && (l.flags & LIST_BITMAP));
            if (dense_cover) {
                cq.seq_tiles = 1;
                for (auto& l : cq.lists)
                    if (l.flags & LIST_BITMAP) l.flags &= ~uint32_t(LIST_COVER);
            }
        }

        // ---- tiling: tile width from the LDS budget and the cover density; spans from the work volume
        const uint32_t L = std::max<uint32_t>(uint32_t(cq.lists.size()), 1);
        const uint64_t range = uint64_t(idx.doc_hi) - idx.doc_lo;
        static const uint32_t ww_max = [] {  // tuning knobs (experiments): VQ_TILE_WORDS_MAX, VQ_TILE_LDS_KB, VQ_SPAN_POSTINGS
            const char* e = std::getenv("VQ_TILE_WORDS_MAX");
            uint32_t v = e ? uint32_t(std::atoi(e)) : 256u;
            uint32_t p2 = 64;
            while (p2 < v && p2 < 2048) p2 <<= 1;
            return p2;
        }();
        static const size_t var_budget = [] {
            const char* e = std::getenv("VQ_TILE_LDS_KB");
            return size_t(e ? std::atoi(e) : 12) * 1024;
        }();
        // postings per span.  Every span warms its own threshold up (until it adopts the query's shared one), so long spans prune
        // better; short spans fill the chip more evenly.  Measured optimum (100 M docs, 256-query launches): 256 Ki postings, 128 Ki
        // for the rich simple kernel and plain simple ANDs (no threshold-driven pruning to warm up; the tail of the launch weighs more).  VQ_SPAN_POSTINGS overrides.
        static const uint64_t span_env = [] {
            const char* e = std::getenv("VQ_SPAN_POSTINGS");
            return uint64_t(e ? std::atoll(e) : 0);
        }();
        const bool and_like = ((cq.simple_flags >> 18) & 1u) || (cq.simple_flags && cq.simple_n > 1 && cq.ops.back().kind == OP_AND);  // rich, or a plain simple AND
        const bool wide_like = (cq.simple_flags >> 24) & 1u;  // k_scan_wide: its count-class pruning gains most from a long warm-up (OR over 8 terms: 27.6 k q/s at 256 Ki, 29.0 k at 512 Ki, 28.7 k at 1 Mi)
        const bool probe_like = (cq.simple_flags >> 25) & 1u;  // k_scan_probe prunes by the query's shared threshold: long spans warm up once (launches of 512: 4.29 ms at 128 Ki, 3.77 ms at 1 Mi)
        // (on a small shard the same number of spans per query is kept — a 1/8 shard with 1 Mi-posting spans would leave half the chip without a wave)
        const uint64_t probe_span = std::min<uint64_t>(std::max<uint64_t>(range / 96, 131072), 1048576);
        // a plain simple OR prunes by its threshold too: on a large shard longer spans warm up once (3-term OR on 100 M docs, launches of 1024: 9.99 ms at 256 Ki,
        // 9.57 at 512 Ki, 9.64 at 1 Mi, 10.86 at 2 Mi); a small shard keeps the span count that fills the chip
        const bool or_like = cq.simple_flags && !((cq.simple_flags >> 18) & 1u) && !wide_like && !probe_like && cq.simple_n > 1 && cq.ops.back().kind == OP_OR;
        const uint64_t span_postings = span_env ? span_env : (probe_like ? probe_span : and_like ? 131072 : wide_like ? 589824 : (or_like && range >= 50'000'000ull) ? 524288 : 262144);  // (wide, launches of 512: 29.7-29.8 k requests/s at 512 Ki, 29.8-30.1 k at 576 Ki, 29.8 k at 608 Ki; AND of two 4-term ORs 30.7 -> 31.3 k)
        uint32_t ww = ww_max;  // W = 32 * ww docs
        const size_t TL = size_t(L) + cq.n_temps;
        while (ww > 64 && (size_t(ww) + TL * ww + size_t(L) * ww / 2) * 4 > var_budget) ww >>= 1;
        if ((size_t(ww) + TL * ww + size_t(L) * ww / 2) * 4 > 96 * 1024) unsupported("too many lists for the LDS tile");
        const uint64_t cover_len = std::max<uint64_t>(root.cover_len, 1);
        // a sparse cover visits about one tile per cover doc: shrink the tile until it holds ~1 cover entry,
        // so that the per-tile clear / prefix work stays proportional to what is actually read
        if (cover_len * (uint64_t(ww) << 5) / std::max<uint64_t>(range, 1) < 64) {
            const uint64_t want = std::max<uint64_t>(range / cover_len, 2048);
            while (ww > 64 && (uint64_t(ww) << 5) / 2 >= want) ww >>= 1;
        }
        if (cq.simple_flags) ww = 256;
        cq.tile_words = ww;
        cq.stack_depth = std::max<uint32_t>(max_depth, 1);
        uint64_t spans = (cq.total_len + span_postings - 1) / span_postings;
        static const uint64_t facet_span = [] {
            const char* e = std::getenv("VQ_FACET_SPAN_POSTINGS");
            return uint64_t(e ? std::atoll(e) : 4096);
        }();
        if (!cq.facets.empty()) spans = (cq.total_len + facet_span - 1) / facet_span;  // every hit walks its facet rows: per-hit work, not per-posting streaming
        const uint64_t tiles = std::max<uint64_t>((range + (uint64_t(ww) << 5) - 1) / (uint64_t(ww) << 5), 1);
        {  // k_tile_scan pays a latency-bound round trip per visited tile (score gathers, facet rows): at most ~16 visited tiles per span
            uint64_t cover_len = 0;
            for (auto& l : cq.lists)
                if (l.flags & LIST_COVER) cover_len += l.len;
            const uint64_t visited = std::max<uint64_t>(std::min<uint64_t>(cover_len, tiles), 1);
            // (k_scan_simple batches its gathers across tiles; tiles with a lot of postings are bandwidth-, not latency-bound)
            if (!cq.simple_flags && cq.total_len / visited < 1024) spans = std::max<uint64_t>(spans, visited / 16);
            if (((cq.simple_flags >> 18) & 1u) && cq.total_len / visited < 1024) spans = std::max<uint64_t>(spans, visited / 64);  // rich: 16384-doc tiles, cheaper each
        }
        if ((cq.simple_flags >> 26) & 1u) {
            // k_scan_ring: a persistent grid draws (query, span) items from a counter, span 0 of every query first: a span needs no length to
            // warm its threshold up (the query's pool is warm after the first round) — short spans even out the end of the launch
            static const uint64_t ring_tiles = [] {
                const char* e = std::getenv("VQ_RING_SPAN_TILES");
                return uint64_t(e ? std::max(1, std::atoi(e)) : 64);
            }();
            const uint64_t ptiles = std::max<uint64_t>((range + (1u << kProbeTileShift) - 1) >> kProbeTileShift, 1);
            spans = (ptiles + ring_tiles - 1) / ring_tiles;
        }
        spans = std::min<uint64_t>(spans, tiles);
        spans = std::min<uint64_t>(std::max<uint64_t>(spans, 1), 4096);
        cq.n_spans = uint32_t(spans);
        if ((cq.simple_flags >> 19) & 1u) {  // spans are slices of the list's ENTRIES: even work whatever the doc distribution
            const uint64_t len = cq.lists[cq.ops[0].list_begin].len;
            cq.n_spans = uint32_t(std::min<uint64_t>(std::max<uint64_t>(len / (cq.facets.empty() ? 16384 : 4096), 1), 4096));
        }
        cq.max_spans = uint32_t(std::min<uint64_t>(std::max<uint64_t>(tiles, 1), 4096));
        if ((cq.simple_flags >> 19) & 1u) cq.max_spans = uint32_t(std::min<uint64_t>(std::max<uint64_t>(cq.lists[cq.ops[0].list_begin].len / 64, 1), 4096));
        compute_layout_bytes(root.cover_len);
    }

    // Bytes this data layout has to move for the query, as far as they are known before it runs (KernelProfile::layout_bytes; the
    // per-hit gathers are counted by the kernels): a list read as a bitmap image costs its words over the visited tiles plus one rank
    // directory entry per tile, a scattered list 4 B per id, a streamed posting list 6 B per posting, a materialised leaf 8 B per entry.
    void compute_layout_bytes(uint64_t cover_len) {
        const uint64_t range = uint64_t(idx.doc_hi) - idx.doc_lo;
        const bool simple = cq.simple_flags != 0;
        const bool wide = (cq.simple_flags >> 24) & 1u;
        const uint64_t tile_docs = wide ? 8192 : ((cq.simple_flags >> 25) & 1u) ? (1u << kProbeTileShift) : simple ? 16384 : uint64_t(cq.tile_words) << 5;
        const uint64_t tiles = std::max<uint64_t>((range + tile_docs - 1) / tile_docs, 1);
        const bool seq = wide ? cq.wide.seq != 0 : simple ? ((cq.simple_flags >> 16) & 1u) : cq.seq_tiles != 0;
        const uint64_t visited = seq ? tiles : std::min<uint64_t>(std::max<uint64_t>(cover_len, 1), tiles);
        auto bitmap_cost = [&]() { return visited * (tile_docs / 8 + 4); };
        uint64_t b = 8ull * cq.top_k;
        if ((cq.simple_flags >> 19) & 1u) b += 8ull * cq.lists[cq.ops[0].list_begin].len;  // k_scan_leaf_f32
        else if (simple && !((cq.simple_flags >> 18) & 1u) && cq.simple_n == 1 && std::getenv("VQ_NO_UNION") == nullptr)
            b += (((cq.simple_flags >> 28) & 1u) ? 4ull : 6ull) * cq.lists[cq.ops[0].list_begin].len;  // k_scan_union: ids and scores streamed, or the tile-packed words
        else if (wide) {
            for (uint32_t k = 0; k < cq.wide.n_leaves; ++k) b += ((cq.wide.bitmap_mask >> k) & 1u) ? bitmap_cost() : 4ull * cq.lists[cq.wide.leaf_list[k]].len;
        } else if ((cq.simple_flags >> 25) & 1u) {  // k_scan_probe: the cover's tile-packed (id, score) words are streamed, a bitmap operand's tiles come with 64 rank entries, an array operand costs 2 B per posting
            for (uint32_t k = 0; k < cq.simple_n; ++k) {
                const uint64_t len = cq.lists[cq.ops[k].list_begin].len;
                b += ((cq.simple_flags >> k) & 1u) ? visited * (tile_docs / 8 + 4 * (tile_docs >> kRankShift)) : ((cq.simple_flags >> (12 + k)) & 1u) ? 2ull * len : 4ull * len;
            }
        } else if (simple) {
            std::vector<bool> seen(cq.lists.size(), false);
            const bool rich = (cq.simple_flags >> 18) & 1u;
            for (uint32_t k = 0; k < cq.simple_n; ++k) {
                const uint32_t li = rich ? cq.simple2.leaf_list[k] : cq.ops[k].list_begin;
                seen[li] = true;
                b += ((cq.simple_flags >> k) & 1u) ? bitmap_cost() : 4ull * cq.lists[li].len;
            }
            for (uint32_t li = 0; li < cq.lists.size(); ++li)
                if (!seen[li]) b += 4ull * cq.lists[li].len;  // side lists
        } else {
            for (auto& l : cq.lists) {
                const bool as_bitmap = (l.flags & LIST_BITMAP) && !(l.flags & LIST_COVER);
                b += as_bitmap ? bitmap_cost() : 4ull * l.len;
            }
        }
        for (auto& f : cq.facets) b += 4ull * f.num_values;
        cq.layout_bytes = b;
    }
};

}  // namespace

float default_score_for_distance_host(uint8_t distance, bool prefix_matches) { return default_score_for_distance(distance, prefix_matches); }
size_t debug_sort_unique(uint32_t* ids, size_t n) {  // (tests: the three regimes of sort_unique_u32 against an independent sort)
    std::vector<uint32_t> v(ids, ids + n);
    sort_unique_u32(v);
    std::copy(v.begin(), v.end(), ids);
    return v.size();
}

// suggest (search_field.rs:194-219): one part's matched terms with lower-cased texts and scores — get_term_ids_in_field with get_scores,
// return_term and return_term_lowercase; the dictionary scan of a fuzzy / prefix part has run on the device (`fuzzy`).
std::vector<SuggestEntry> suggest_part(const Index& idx, const RequestSearchPart& part, const FuzzyTable* fuzzy) {
    Request dummy;
    Compiler c(idx, dummy, fuzzy);
    Leaf l;
    l.part = &part;
    l.get_scores = true;
    l.return_term = true;
    l.return_term_lowercase = true;
    c.lookup_terms(idx, l, true, false);
    if (part.token_value) c.apply_token_value(*part.token_value, l);
    std::vector<SuggestEntry> out;
    std::map<uint32_t, const std::string*> text;
    for (auto& t : l.terms) text[t.first] = &t.second;
    for (auto& h : l.hits_scores) out.push_back(SuggestEntry{*text.at(h.first), h.second, h.first});
    return out;
}


// highlight (search_field.rs:233-245): the texts that contain a matched token, each with its snippet (highlight_field.rs:187-272) and the
// best score among its matched tokens (resolve_token_hits_to_text_id with snippets, search_field.rs:550-639).  `part.terms` are already
// normalised; the dictionary scan of a fuzzy / prefix part has run on the device (`fuzzy`).  Host work over the host copies of
// tokens_to_text_id / text_id_to_token_ids: one text at a time, a few dozen tokens each.
namespace {
// the snippet of n tokens: is_hit(i) marks the tokens to tag, append(out, i) writes token i's text
template <class IsHit, class Append>
std::string snippet_of_tokens(size_t n, IsHit is_hit, Append append_text, const vqreq::SnippetInfo& opt, bool* any) {
    const int64_t around = opt.num_words_around_snippet * 2;  // token separator token separator
    // one walk: hits closer than `around` tokens share a window (group_hit_positions_for_snippet :19-37); a window reaches `around` tokens
    // to both sides of its hits (grouped_to_positions_for_snippet :39-43)
    std::string out;
    int64_t first_hit = -1, last_hit = -1, group_first = -1, group_last = -1;
    uint64_t windows = 0;
    auto close_group = [&]() {
        if (group_first < 0) return;
        if (windows < opt.max_snippets) {
            if (windows) out += opt.snippet_connector;
            const size_t lo = size_t(std::max<int64_t>(group_first - around, 0)), hi = size_t(std::min<int64_t>(group_last + around + 1, int64_t(n)));
            for (size_t i = lo; i < hi; ++i) {
                if (is_hit(i)) {
                    out += opt.snippet_start_tag;
                    append_text(out, i);
                    out += opt.snippet_end_tag;
                } else append_text(out, i);
            }
        }
        ++windows;
    };
    for (size_t i = 0; i < n; ++i) {
        if (!is_hit(i)) continue;
        if (first_hit < 0) first_hit = int64_t(i);
        if (group_first < 0 || int64_t(i) - group_last >= around) {
            close_group();
            group_first = int64_t(i);
        }
        group_last = last_hit = int64_t(i);
    }
    close_group();
    *any = first_hit >= 0;
    if (!*any) return out;
    if (first_hit > around) out.insert(0, opt.snippet_connector);             // ellipsis_snippet :73-90
    if (last_hit < int64_t(n) - around) out += opt.snippet_connector;
    return out;
}
std::string snippet_of_text(const Dictionary& dict, const uint32_t* toks, size_t n, const std::vector<uint32_t>& wanted_sorted, const vqreq::SnippetInfo& opt, bool* any) {
    return snippet_of_tokens(
        n, [&](size_t i) { return std::binary_search(wanted_sorted.begin(), wanted_sorted.end(), toks[i]); },
        [&](std::string& out, size_t i) {
            if (toks[i] < dict.terms.size()) out += dict.terms[toks[i]];
        },
        opt, any);
}
void check_snippet_window(const vqreq::SnippetInfo& opt) {
    if (opt.num_words_around_snippet < 0 || opt.num_words_around_snippet > 0x3FFFFFFF)  // the reference's window arithmetic overflows / panics there
        throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"snippet_info.num_words_around_snippet out of range\" ");
}
}  // namespace

// highlight_text (highlight_field.rs:92-146): `text` with the tokens that are in `terms` tagged — what why_found highlighting applies to every text
// of a returned document (highlight_on_original_document, :148-185) with the field's why_found_terms.  nullopt: nothing to highlight.
std::optional<std::string> highlight_text(const std::string& text, const std::vector<std::string>& terms, const vqreq::SnippetInfo& opt, bool tokenized) {
    check_snippet_window(opt);
    std::vector<std::string> set = terms;
    std::sort(set.begin(), set.end());
    set.erase(std::unique(set.begin(), set.end()), set.end());
    if (set.size() == 1 && set[0] == text) return opt.snippet_start_tag + text + opt.snippet_end_tag;  // one hit that is the whole text
    if (!tokenized) return std::nullopt;
    const std::vector<vqtext::TokenSpan> tokens = vqtext::tokenize_grouped(text);
    std::vector<char> hit(tokens.size(), 0);
    for (size_t i = 0; i < tokens.size(); ++i) hit[i] = std::binary_search(set.begin(), set.end(), text.substr(tokens[i].begin, tokens[i].end - tokens[i].begin));
    bool any = false;
    std::string out = snippet_of_tokens(
        tokens.size(), [&](size_t i) { return hit[i] != 0; }, [&](std::string& o, size_t i) { o.append(text, tokens[i].begin, tokens[i].end - tokens[i].begin); }, opt, &any);
    // (contains_any_token is set while the windows are written: with max_snippets == 0 nothing is)
    if (!any || opt.max_snippets == 0) return std::nullopt;
    return out;
}

std::vector<SuggestEntry> highlight_part(const Index& idx, const RequestSearchPart& part, const FuzzyTable* fuzzy) {
    static const vqreq::SnippetInfo kDefault;
    const vqreq::SnippetInfo& opt = part.has_snippet_info ? part.snippet_info : kDefault;
    check_snippet_window(opt);
    RequestSearchPart lookup = part;  // get_term_ids_in_field does not look at the snippet fields
    lookup.snippet.reset();
    lookup.has_snippet_info = false;
    Request dummy;
    Compiler c(idx, dummy, fuzzy);
    Leaf l;
    l.part = &lookup;
    l.get_scores = true;
    c.lookup_terms(idx, l, true, false);
    if (lookup.token_value) c.apply_token_value(*lookup.token_value, l);

    struct TextHit {
        uint32_t text;
        float score;
        uint32_t token, order;
    };
    std::vector<TextHit> hits;
    std::vector<std::pair<uint32_t, float>> ranked;  // (text id, best score): SearchFieldResult::hits_scores after the resolve step
    const std::string& path = l.path;
    auto cit = idx.columns.find(path.substr(0, path.size() - std::strlen(TEXTINDEX)));
    const bool tokenized = cit != idx.columns.end() && cit->second.tokenize;
    const bool add_snippets = part.snippet.value_or(false);
    std::map<uint32_t, std::string> snippets;
    for (auto& h : l.hits_scores) ranked.push_back(h);
    if (tokenized) {
        auto kit = idx.kv.find(path + TOKENS_TO_TEXT_ID);
        if (kit == idx.kv.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path + TOKENS_TO_TEXT_ID);
        uint32_t order = 0;
        for (auto& h : l.hits_scores) {
            const uint32_t *b, *e;
            if (!kit->second.host_row(h.first, &b, &e)) continue;
            for (const uint32_t* v = b; v != e; ++v) hits.push_back({*v, h.second, h.first, order++});
        }
        std::sort(hits.begin(), hits.end(), [](const TextHit& a, const TextHit& b) { return a.text != b.text ? a.text < b.text : a.order < b.order; });
        if (!hits.empty()) {
            if (add_snippets) ranked.clear();  // :608-610
            const KVStore* t2t = nullptr;
            const Dictionary& dict = idx.dict.at(path);
            std::vector<uint32_t> wanted;
            for (size_t i = 0; i < hits.size();) {
                size_t j = i;
                float best = hits[i].score;
                wanted.clear();
                for (; j < hits.size() && hits[j].text == hits[i].text; ++j) {
                    if (std::fabs(hits[j].score) >= std::fabs(best)) best = hits[j].score;  // Iterator::max_by_key keeps the last maximum
                    wanted.push_back(hits[j].token);
                }
                const uint32_t text = hits[i].text;
                ranked.push_back({text, best});
                if (add_snippets) {
                    if (!t2t) {
                        auto tit = idx.kv.find(path + ".text_id_to_token_ids");
                        if (tit == idx.kv.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path + ".text_id_to_token_ids");
                        t2t = &tit->second;
                    }
                    std::sort(wanted.begin(), wanted.end());
                    const uint32_t *b, *e;
                    if (t2t->host_row(text, &b, &e)) {
                        bool any = false;
                        std::string sn = snippet_of_text(dict, b, size_t(e - b), wanted, opt, &any);
                        if (any) snippets[text] = std::move(sn);
                    } else if (std::binary_search(wanted.begin(), wanted.end(), text))  // the text is its own only token: all of it (highlight_field.rs:198-203)
                        snippets[text] = opt.snippet_start_tag + (text < dict.terms.size() ? dict.terms[text] : std::string()) + opt.snippet_end_tag;
                }
                i = j;
            }
        }
    }
    std::vector<SuggestEntry> out;
    for (auto& r : ranked) {  // get_text_score_id_from_result(false, ..) search_field.rs:160-192: indexing `highlight` panics for a hit without a snippet
        auto it = snippets.find(r.first);
        if (it == snippets.end())
            throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"highlight: hit " + std::to_string(r.first) + " has no snippet (the reference panics)\" ");
        out.push_back(SuggestEntry{it->second, r.second, r.first});
    }
    return out;
}

// get_why_found (src/search/why_found.rs:11-50) for the finished requests that asked for why_found together with select: for every searched field
// and every returned anchor, the anchor's texts of the field (join_anchor_to_leaf, facet.rs:75-93) highlighted with all term ids the search matched
// there (highlight_document with DEFAULT_SNIPPETINFO, highlight_field.rs:187-272); a text without a hit leaves no entry.
void complete_why_found_requests(const Index& idx, std::vector<std::unique_ptr<Result>>& results, std::vector<int>& status, std::vector<std::string>& errors) {
    static const vqreq::SnippetInfo kDefault;
    for (size_t i = 0; i < results.size(); ++i) {
        if (status[i] != 0 || !results[i] || !results[i]->why_found_plan) continue;
        Result& R = *results[i];
        try {
            for (auto& [path, wanted] : *R.why_found_plan) {  // (wanted: sorted, unique)
                if (wanted.empty() || R.ids.empty()) continue;  // why_found.rs:29-31
                const std::string field_name = path.substr(0, path.size() - std::strlen(TEXTINDEX));
                const std::vector<std::string> steps = get_steps_to_anchor(field_name);
                auto store = [&](const std::string& name) -> const KVStore& {
                    auto it = idx.kv.find(name);
                    if (it == idx.kv.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + name);
                    return it->second;
                };
                std::vector<const KVStore*> chain;
                for (auto& st : steps) chain.push_back(&store(st + PARENT_TO_VALUE_ID));
                const KVStore& t2t = store(steps.back() + ".text_id_to_token_ids");
                auto dit = idx.dict.find(steps.back());
                if (dit == idx.dict.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + steps.back() + ".fst");
                const Dictionary& dict = dit->second;
                std::vector<uint32_t> level, next;
                for (uint32_t anchor : R.ids) {
                    level.assign(1, anchor);
                    for (const KVStore* st : chain) {
                        next.clear();
                        for (uint32_t id : level) {
                            const uint32_t *rb, *re;
                            if (st->host_row(id, &rb, &re)) next.insert(next.end(), rb, re);
                        }
                        level.swap(next);
                    }
                    for (uint32_t value_id : level) {
                        const uint32_t *b, *e;
                        if (t2t.host_row(value_id, &b, &e)) {
                            bool any = false;
                            std::string sn = snippet_of_text(dict, b, size_t(e - b), wanted, kDefault, &any);
                            if (any) R.why_found_info[anchor][field_name].push_back(std::move(sn));
                        } else if (std::binary_search(wanted.begin(), wanted.end(), value_id))  // the text is its own only token: all of it (highlight_field.rs:198-203)
                            R.why_found_info[anchor][field_name].push_back(kDefault.snippet_start_tag + (value_id < dict.terms.size() ? dict.terms[value_id] : std::string()) +
                                                                           kDefault.snippet_end_tag);
                    }
                }
            }
        } catch (const VelociError& e) {
            status[i] = e.code;
            errors[i] = e.what();
            results[i].reset();
        }
    }
}

// ---- dictionary scans requested by a batch (collected before compilation, answered by k_dict_scan)
std::string fuzzy_key(const RequestSearchPart& p) {
    std::string path = p.path;
    if (!ends_with(path, TEXTINDEX)) path += TEXTINDEX;
    std::string k;
    key_s(k, path);
    key_s(k, p.terms.empty() ? std::string() : p.terms[0]);
    k += std::to_string(clamped_lev(p)) + (p.starts_with ? "p" : "-") + (p.ignore_case ? (*p.ignore_case ? "T" : "F") : "N");
    return k;
}
bool needs_dictionary_scan(const RequestSearchPart& p) { return !p.terms.empty() && !p.is_regex && (clamped_lev(p) != 0 || p.starts_with); }

static void probe_part(const Index& idx, const RequestSearchPart& p, FuzzyTable& table) {
    if (!needs_dictionary_scan(p)) return;
    const std::string key = fuzzy_key(p);
    if (table.count(key)) return;
    FuzzyProbe fp;
    fp.key = key;
    fp.path = p.path;
    if (!ends_with(fp.path, TEXTINDEX)) fp.path += TEXTINDEX;
    auto dit = idx.dict.find(fp.path);
    if (dit == idx.dict.end()) return;  // the compiler reports FstNotFound
    // match-set automaton (search_field.rs:85-95): built from the ORIGINAL term
    fp.max_d = std::min<uint32_t>(clamped_lev(p), 4);
    fp.transposition = p.ignore_case.value_or(false);
    fp.ci = p.ignore_case.value_or(true);
    fp.prefix = p.starts_with;
    fp.lower_term = vqtext::to_lower_utf8(p.terms[0]);
    fp.lev = clamped_lev(p);
    fp.check_prefix = p.starts_with || fp.lev != 0;  // :302
    const auto cps = vqtext::decode_utf8(p.terms[0]);
    if (!dit->second.bmp_only) {
        fp.status = ERR_UNSUPPORTED;
        fp.error = "fuzzy / prefix search on " + fp.path + ": dictionary holds code points above U+FFFF";
    } else if (cps.size() > 64) {
        fp.status = ERR_UNSUPPORTED;
        fp.error = "fuzzy / prefix search with a term longer than 64 characters";
    }
    for (uint32_t cp : cps) {
        if (cp > 0xFFFFu && fp.status == 0) {
            fp.status = ERR_UNSUPPORTED;
            fp.error = "fuzzy / prefix search with a code point above U+FFFF";
        }
        fp.query.push_back(uint16_t(fp.ci ? vqtext::lower_cp(cp) : cp));
    }
    table.emplace(key, std::move(fp));
}
static void probe_tree(const Index& idx, const SearchRequest& r, FuzzyTable& table) {
    if (r.kind == SearchRequest::Search) probe_part(idx, r.part, table);
    else
        for (auto& q : r.tree.queries) probe_tree(idx, q, table);
}
// score of every match of a probe (search_field.rs:304-321), shared by all requests of the batch that contain the leaf
void score_fuzzy_probe(const Index& idx, FuzzyProbe& fp) {
    if (fp.status != 0) return;
    const Dictionary& dict = idx.dict.at(fp.path);
    fp.scores.resize(fp.matches.size());
    for (size_t i = 0; i < fp.matches.size(); ++i) {
        const std::string lower_hit = vqtext::to_lower_utf8(dict.terms[fp.matches[i]]);
        const bool prefix_matches = fp.check_prefix && lower_hit.size() >= fp.lower_term.size() && lower_hit.compare(0, fp.lower_term.size(), fp.lower_term) == 0;
        fp.scores[i] = default_score_for_distance(scoring_distance(lower_hit, fp.lower_term, fp.lev), prefix_matches);
    }
}

void collect_suggest_probes(const Index& idx, const Request& req, FuzzyTable& table) {
    if (req.suggest)
        for (auto& p : *req.suggest) probe_part(idx, p, table);
}

void collect_fuzzy_probes(const Index& idx, const Request& req, FuzzyTable& table) {
    if (req.search_req) probe_tree(idx, *req.search_req, table);
    if (req.filter) probe_tree(idx, *req.filter, table);
    if (req.phrase_boosts)
        for (auto& pb : *req.phrase_boosts) {
            probe_part(idx, pb.search1, table);
            probe_part(idx, pb.search2, table);
        }
    if (req.boost_term)
        for (auto& p : *req.boost_term) probe_part(idx, p, table);
}

// request.explain goes into the options of every part before the parts are collected — it is part of their equality (execution_plan.rs:46-106)
static void propagate_explain(Request& request) {
    auto merge_explain = [](std::optional<SearchRequestOptions>& o) {
        if (!o) o = SearchRequestOptions{};
        o->explain = true;
    };
    std::function<void(SearchRequest&)> walk = [&](SearchRequest& r) {
        if (r.kind == SearchRequest::Search) merge_explain(r.part.options);
        else {
            merge_explain(r.tree.options);
            for (auto& q : r.tree.queries) walk(q);
        }
    };
    if (request.phrase_boosts)
        for (auto& el : *request.phrase_boosts) {
            merge_explain(el.search1.options);
            merge_explain(el.search2.options);
        }
    if (request.search_req) walk(*request.search_req);
    if (request.filter) walk(*request.filter);
}
bool request_wants_explain(const vqreq::Request& req) {
    if (req.explain) return true;
    bool any = false;
    std::function<void(const SearchRequest&)> walk = [&](const SearchRequest& r) {
        if (r.kind == SearchRequest::Search) any = any || part_explains(r.part);
        else
            for (auto& q : r.tree.queries) walk(q);
    };
    if (req.search_req) walk(*req.search_req);
    return any;
}

CompiledQuery compile_query(const Index& idx, const vqreq::Request& req_in, const FuzzyTable* fuzzy, const UnionTable* unions, const QueryCounts* counts,
                            const RangeTable* ranges, Boost1nCache* boost_cache, const LocalityTable* localities) {
    std::unique_ptr<Request> explained;
    if (req_in.explain) {
        explained = std::make_unique<Request>(req_in);
        propagate_explain(*explained);
    }
    const vqreq::Request& req = explained ? *explained : req_in;
    PhaseTimer pt(5);
    Compiler c(idx, req, fuzzy);
    c.boost_cache = boost_cache;
    c.localities = localities;
    c.unions = unions;
    c.counts = counts;
    c.ranges = ranges;
    try {
        c.run();
        if (!c.cq.range_requests.empty()) {
            c.cq.status = kStatusNeedsRanges;
            c.cq.error = "internal: range jobs pending";
        } else if (!c.cq.union_requests.empty() || !c.cq.locality_requests.empty() || !c.cq.boost1n_requests.empty()) {
            c.cq.status = kStatusNeedsUnion;
            c.cq.error = "internal: union jobs pending";
        } else if (c.cq.n_counts) {
            c.cq.status = kStatusNeedsCounts;
            c.cq.error = "internal: count pre-pass pending";
        }
    } catch (const VelociError& e) {
        c.cq.status = e.code;
        c.cq.error = e.what();
    }
    return std::move(c.cq);
}

}  // namespace vq
