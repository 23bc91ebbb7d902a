// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See veloci_oracle.hpp for the rules.
// CPU restatement of the reference's query path; citations are /root/reference paths.
#include "veloci_oracle.hpp"

#include <chrono>
#include <regex>
#include <cstdio>

namespace vo {

// =====================================================================================
// Index accessors (src/persistence.rs:430-470)
// =====================================================================================
const TokenToAnchorScore& Index::get_token_to_anchor(const std::string& path) const {
    std::string p = path + TO_ANCHOR_ID_SCORE;
    auto it = token_to_anchor_score.find(p);
    if (it == token_to_anchor_score.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + p);
    return it->second;
}
const PhrasePairToAnchor& Index::get_phrase_pair_to_anchor(const std::string& path) const {
    auto it = phrase_pair_to_anchor.find(path);
    if (it == phrase_pair_to_anchor.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
    return it->second;
}
const BoostStore& Index::get_boost(const std::string& path) const {
    auto it = boost_valueid_to_value.find(path);
    if (it == boost_valueid_to_value.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
    return it->second;
}

static void push_boost_record(ExplainMap* explain, uint32_t id, float v) {
    if (!explain) return;
    Explain e;
    e.kind = Explain::Boost;
    e.a = v;
    (*explain)[id].push_back(e);
}
std::string explain_json(const std::vector<Explain>& records) {
    auto f = [](float v) {
        char buf[48];
        std::snprintf(buf, sizeof buf, "%.9g", double(v));
        return std::string(buf);
    };
    std::string out = "[";
    for (size_t i = 0; i < records.size(); ++i) {
        const Explain& e = records[i];
        if (i) out += ",";
        switch (e.kind) {
            case Explain::Boost: out += "{\"Boost\":" + f(e.a) + "}"; break;
            case Explain::MaxTokenToTextId: out += "{\"MaxTokenToTextId\":" + f(e.a) + "}"; break;
            case Explain::OrSumOverDistinctTerms: out += "{\"OrSumOverDistinctTerms\":" + f(e.a) + "}"; break;
            case Explain::TermToAnchor:
                out += "{\"TermToAnchor\":{\"term_score\":" + f(e.a) + ",\"anchor_score\":" + f(e.b) + ",\"final_score\":" + f(e.c) + ",\"term_id\":" + std::to_string(e.term_id) + "}}";
                break;
            case Explain::LevenshteinScore:
                out += "{\"LevenshteinScore\":{\"score\":" + f(e.a) + ",\"text_or_token_id\":";
                vqjson::escape_to(out, e.text);
                out += ",\"term_id\":" + std::to_string(e.term_id) + "}}";
                break;
        }
    }
    return out + "]";
}

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && std::memcmp(s.data() + s.size() - n, suf, n) == 0;
}
// util.rs:131-137 extract_field_name: drop the trailing ".textindex" (10 chars)
static std::string extract_field_name(const std::string& field) {
    auto cps = vqtext::decode_utf8(field);
    size_t keep = cps.size() >= 10 ? cps.size() - 10 : 0;
    std::string out;
    for (size_t i = 0; i < keep; ++i) vqtext::append_utf8(out, cps[i]);
    return out;
}
static bool is_anchor_identity(const Index& idx, const std::string& textindex_path) {
    auto it = idx.columns.find(extract_field_name(textindex_path));
    return it != idx.columns.end() && it->second.is_anchor_identity_column;
}

// =====================================================================================
// itertools 0.12 kmerge_by (third party; restated: binary heap of (head, tail), sift_down)
// =====================================================================================
template <class T, class Less>
class KMerge {
    struct HeadTail {
        T head;
        const T* cur;
        const T* end;
    };
    std::vector<HeadTail> heap_;
    Less less_;
    bool lt(const HeadTail& a, const HeadTail& b) { return less_(a.head, b.head); }
    void sift_down(size_t index) {
        size_t pos = index;
        size_t child = 2 * pos + 1;
        while (child + 1 < heap_.size()) {
            child += lt(heap_[child + 1], heap_[child]) ? 1 : 0;
            if (!lt(heap_[child], heap_[pos])) return;
            std::swap(heap_[pos], heap_[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child + 1 == heap_.size() && lt(heap_[child], heap_[pos])) std::swap(heap_[pos], heap_[child]);
    }

public:
    KMerge(const std::vector<std::pair<const T*, const T*>>& sources, Less less) : less_(less) {
        for (auto& s : sources)
            if (s.first != s.second) heap_.push_back({*s.first, s.first + 1, s.second});
        for (size_t i = heap_.size() / 2; i-- > 0;) sift_down(i);
    }
    bool next(T& out) {
        if (heap_.empty()) return false;
        HeadTail& top = heap_[0];
        out = top.head;
        if (top.cur != top.end) {
            top.head = *top.cur++;
        } else {
            heap_[0] = heap_.back();  // swap_remove(0)
            heap_.pop_back();
        }
        sift_down(0);
        return true;
    }
};

// =====================================================================================
// A1 — dictionary lookup (src/search/search_field.rs:27-33, 68-99, 277-398, 691-732)
// =====================================================================================
float get_default_score_for_distance(uint8_t distance, bool prefix_matches) {
    if (prefix_matches) return 2.0f / (std::log2(float(distance) + 1.0f) + 0.2f);
    return 2.0f / (float(distance) + 0.2f);
}

// search_field.rs:705-732 (plain Levenshtein over chars, u8 arithmetic, strings < 255 bytes)
uint8_t distance(const std::string& s1, const std::string& s2) {
    if (s1.size() >= 255 || s2.size() >= 255) return 255;
    auto c1 = vqtext::decode_utf8(s1), c2 = vqtext::decode_utf8(s2);
    size_t len_s1 = c1.size();
    uint8_t column[256];
    std::memset(column, 0, sizeof column);
    for (size_t i = 0; i <= len_s1; ++i) column[i] = uint8_t(i);
    for (size_t x = 0; x < c2.size(); ++x) {
        column[0] = uint8_t(x + 1);
        uint8_t lastdiag = uint8_t(x);
        for (size_t y = 0; y < c1.size(); ++y) {
            if (c1[y] != c2[x]) lastdiag = uint8_t(lastdiag + 1);
            uint8_t olddiag = column[y + 1];
            column[y + 1] = std::min<uint8_t>(uint8_t(column[y + 1] + 1), std::min<uint8_t>(uint8_t(column[y] + 1), lastdiag));
            lastdiag = olddiag;
        }
    }
    return column[len_s1];
}

static inline bool cp_eq(uint32_t a, uint32_t b, bool ci) { return a == b || (ci && vqtext::lower_cp(a) == vqtext::lower_cp(b)); }

// veloci_levenshtein_automata (third party, restated): Levenshtein distance, optionally with adjacent
// transposition at cost one (optimal string alignment), optionally case-insensitive.
static uint32_t lev_span(const uint32_t* a, size_t n, const uint32_t* b, size_t m, bool transposition, bool ci, bool prefix_min) {
    // rows: a (dictionary term), columns: b (query).  prefix_min: min over all prefixes of a.
    uint32_t buf[3][260];
    std::vector<uint32_t> big;
    uint32_t *prev2 = buf[0], *prev = buf[1], *cur = buf[2];
    if (m + 1 > 260) {
        big.resize(3 * (m + 1));
        prev2 = big.data();
        prev = prev2 + (m + 1);
        cur = prev + (m + 1);
    }
    for (size_t j = 0; j <= m; ++j) prev[j] = uint32_t(j);
    uint32_t best = prev[m];
    for (size_t i = 1; i <= n; ++i) {
        cur[0] = uint32_t(i);
        for (size_t j = 1; j <= m; ++j) {
            uint32_t cost = cp_eq(a[i - 1], b[j - 1], ci) ? 0 : 1;
            uint32_t v = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + cost);
            if (transposition && i > 1 && j > 1 && cp_eq(a[i - 1], b[j - 2], ci) && cp_eq(a[i - 2], b[j - 1], ci))
                v = std::min(v, prev2[j - 2] + 1);
            cur[j] = v;
        }
        best = std::min(best, cur[m]);
        uint32_t* t = prev2;
        prev2 = prev;
        prev = cur;
        cur = t;
    }
    return prefix_min ? best : prev[m];
}
uint32_t levenshtein_cps(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b, bool transposition, bool ci) {
    return lev_span(a.data(), a.size(), b.data(), b.size(), transposition, ci, false);
}
// starts_with: `Automaton::starts_with()` (fst 0.4.7 automaton/mod.rs StartsWith) accepts once ANY prefix of the
// dictionary term is accepted == lev_span(..., prefix_min = true) <= max distance.

// search_field.rs:691-702 distance_dfa: the DFA (transposition cost one, built over lower_term, case
// sensitive) reports Exact(d) for d <= max distance, otherwise the plain `distance` is used.
static uint8_t distance_dfa(const std::string& lower_hit, const std::string& lower_term, uint32_t dfa_max_distance) {
    uint32_t d = levenshtein_cps(vqtext::decode_utf8(lower_hit), vqtext::decode_utf8(lower_term), true, false);
    if (d <= dfa_max_distance) return uint8_t(d);
    return distance(lower_hit, lower_term);
}

// sort.rs:24-34
template <class T, class Cmp, class NewWorst>
static void check_apply_top_n_sort(std::vector<T>& new_data, uint32_t top_n, Cmp cmp_less, NewWorst new_worst) {
    if (!new_data.empty() && new_data.size() == size_t(top_n) + 200) {
        std::sort(new_data.begin(), new_data.end(), cmp_less);
        new_data.resize(top_n);
        new_worst(new_data.back());
    }
}
// search.rs:122-130 sort_by_score_and_id as a strict-weak "a before b"
static inline bool score_id_before(const Hit& a, const Hit& b) {
    if (a.score == b.score) return a.id > b.id;
    return a.score > b.score;
}

// search_field.rs:72-83: regex-automata 0.1.9 dense DFA over the term's bytes, unanchored at the start (the builder's default acts as if the
// pattern began with `(?s:.)*?`), accepted when the walk ENDS in a match state; `starts_with` accepts once any prefix did.  Restated with
// std::regex (ECMAScript grammar, the common subset of the two syntaxes; byte-wise: `.` is one byte here, one scalar value there).
static bool regex_matches(const std::regex& whole, const std::regex& anywhere, bool starts_with, const std::string& term) {
    if (starts_with) return std::regex_search(term, anywhere);
    return std::regex_match(term, whole);
}

SearchFieldResult get_term_ids_in_field(const Index& index, PlanRequestSearchPart& options) {
    RequestSearchPart& req = options.request;
    if (req.terms.empty()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"terms is empty\" ");  // reference: index out of bounds panic
    if (!ends_with(req.path, TEXTINDEX)) req.path += TEXTINDEX;  // :278-280
    SearchFieldResult result;
    result.request = req;

    const std::string lower_term = vqtext::to_lower_utf8(req.terms[0]);  // :284
    const auto lower_term_cps = vqtext::decode_utf8(lower_term);
    if (req.levenshtein_distance) {  // :285-287
        if (lower_term_cps.empty()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"empty term with levenshtein_distance\" ");
        req.levenshtein_distance = std::min<uint32_t>(*req.levenshtein_distance, uint32_t(lower_term_cps.size()) - 1);
    }
    const bool limit_result = req.top.has_value();  // :292
    float worst_score = -std::numeric_limits<float>::max();
    const uint32_t top_n_search = uint32_t(req.top.value_or(10) + req.skip.value_or(0));
    const uint32_t lev = req.levenshtein_distance.value_or(0);
    const bool should_check_prefix_match = req.starts_with || lev != 0;  // :302

    auto fit = index.fst.find(req.path);  // :148-152
    if (fit == index.fst.end()) throw VelociError(ERR_FST_NOT_FOUND, "field does not exist " + req.path + " (fst not found)");
    const Fst& fst = fit->second;

    // match-set automaton, :85-95 — built from the ORIGINAL term
    const uint32_t match_max_d = std::min<uint32_t>(lev, 4);
    const bool match_transposition = req.ignore_case.value_or(false);
    const bool match_ci = req.ignore_case.value_or(true);
    const auto query_cps = vqtext::decode_utf8(req.terms[0]);

    auto callback = [&](const std::string& text_or_token, uint32_t token_text_id) {  // :304-354
        if (options.get_ids) result.hits_ids.push_back(token_text_id);
        if (options.get_scores) {
            std::string line_lower = vqtext::to_lower_utf8(text_or_token);
            bool prefix_matches = should_check_prefix_match && line_lower.size() >= lower_term.size() &&
                                  std::memcmp(line_lower.data(), lower_term.data(), lower_term.size()) == 0;
            float score = get_default_score_for_distance(distance_dfa(line_lower, lower_term, lev), prefix_matches);
            if (limit_result) {
                if (score < worst_score) return;
                check_apply_top_n_sort(result.hits_scores, top_n_search, score_id_before, [&](const Hit& w) { worst_score = w.score; });
            }
            result.hits_scores.push_back(Hit{token_text_id, score});
            if (is_explain(req)) {  // :334-343 (insert: a later hit with the same id would replace the entry)
                Explain e;
                e.kind = Explain::LevenshteinScore;
                e.a = score;
                e.term_id = token_text_id;
                e.text = text_or_token;
                result.explain[token_text_id] = std::vector<Explain>{e};
            }
        }
        if (options.return_term || options.store_term_texts)  // :347-353 (not reached by a hit the top-n cut has just dropped, :324-327)
            result.terms[token_text_id] = options.return_term_lowercase ? vqtext::to_lower_utf8(text_or_token) : text_or_token;
    };

    // The reference walks the FST under the automaton; stream order == bytewise order == id order.
    // Restated as: exact (distance 0, no prefix) -> lookup in the lowercase table built at load time,
    // candidates verified with the same predicate; everything else -> scan of the sorted term table.
    auto matches = [&](const std::string& term) -> bool {
        uint32_t tbuf[256];
        const uint32_t* tp;
        size_t tn;
        std::vector<uint32_t> tcps;
        if (term.size() <= 256 && vqtext::is_ascii(term.data(), term.size())) {
            for (size_t i = 0; i < term.size(); ++i) tbuf[i] = (unsigned char)term[i];
            tp = tbuf;
            tn = term.size();
        } else {
            tcps = vqtext::decode_utf8(term);
            tp = tcps.data();
            tn = tcps.size();
        }
        if (!req.starts_with && (tn > query_cps.size() + match_max_d || tn + match_max_d < query_cps.size())) return false;
        return lev_span(tp, tn, query_cps.data(), query_cps.size(), match_transposition, match_ci, req.starts_with) <= match_max_d;
    };
    if (req.is_regex) {  // :72-83
        std::regex whole, anywhere;
        try {
            const auto flags = std::regex::ECMAScript | (req.ignore_case.value_or(true) ? std::regex::icase : std::regex::ECMAScript);
            whole = std::regex("[\\s\\S]*?(?:" + req.terms[0] + ")", flags);
            anywhere = std::regex(req.terms[0], flags);
        } catch (const std::regex_error& e) {
            throw VelociError(ERR_INVALID_REQUEST, std::string("InvalidRequest: \"regex ") + e.what() + "\" ");  // (the reference unwrap()s: a panic)
        }
        for (uint32_t id = 0; id < fst.terms.size(); ++id)
            if (regex_matches(whole, anywhere, req.starts_with, fst.terms[id])) callback(fst.terms[id], id);
    } else if (match_max_d == 0 && !req.starts_with) {
        std::vector<uint32_t> cand;
        if (match_ci) {
            auto it = fst.lower_map.find(vqtext::to_lower_utf8(req.terms[0]));
            if (it != fst.lower_map.end()) cand = it->second;
        } else {
            auto it = std::lower_bound(fst.terms.begin(), fst.terms.end(), req.terms[0]);
            if (it != fst.terms.end() && *it == req.terms[0]) cand.push_back(uint32_t(it - fst.terms.begin()));
        }
        for (uint32_t id : cand)  // ascending ids
            if (matches(fst.terms[id])) callback(fst.terms[id], id);
    } else {
        for (uint32_t id = 0; id < fst.terms.size(); ++id)
            if (matches(fst.terms[id])) callback(fst.terms[id], id);
    }

    if (req.boost) {  // :359-364
        for (auto& h : result.hits_scores) h.score *= *req.boost;
    }
    if (limit_result) {  // :373-376 (unstable sort: tie order unspecified in the reference; stable here)
        std::stable_sort(result.hits_scores.begin(), result.hits_scores.end(), [](const Hit& a, const Hit& b) { return a.score > b.score; });
        if (result.hits_scores.size() > top_n_search) result.hits_scores.resize(top_n_search);
    }
    if (options.store_term_id_hits && !result.hits_scores.empty()) {  // :379-383
        std::vector<uint32_t> ids;
        for (auto& h : result.hits_scores) ids.push_back(h.id);
        result.term_id_hits_in_field[req.path][req.terms[0]] = ids;
    }
    if (options.store_term_texts && !result.terms.empty()) {  // :386-389
        std::vector<std::string> texts;
        for (auto& kv : result.terms) texts.push_back(kv.second);
        result.term_text_in_field[req.path] = texts;
    }
    if (req.token_value) {  // :391-395
        RequestBoostPart tb = *req.token_value;
        tb.path = tb.path + TEXTINDEX + TOKEN_VALUES;
        req.token_value->path = tb.path;
        add_boost(index, tb, result);
    }
    return result;
}

// =====================================================================================
// A2 — posting decode + score (src/search/search_field.rs:400-504, 540-548)
// =====================================================================================
static inline bool should_filter(const FilterResult* filter, uint32_t id) {  // :540-548
    if (!filter) return false;
    if (!filter->is_set) return false;  // FilterResult::Vec is not applied here
    return filter->set.find(id) == filter->set.end();
}

SearchFieldResult resolve_token_to_anchor(const Index& index, const RequestSearchPart& options_in, const FilterResult* filter,
                                          const SearchFieldResult& result) {
    RequestSearchPart options = options_in;
    if (!ends_with(options.path, TEXTINDEX)) options.path += TEXTINDEX;
    SearchFieldResult res = SearchFieldResult::new_from(result);
    std::vector<Hit> anchor_ids_hits;
    const TokenToAnchorScore& store = index.get_token_to_anchor(options.path);  // :416
    for (const Hit& hit : result.hits_scores) {  // :419-444
        if (uint64_t(hit.id) + 1 >= store.offsets.size()) continue;  // get_score_iter: empty beyond the table
        uint64_t b = store.offsets[hit.id], e = store.offsets[hit.id + 1];
        anchor_ids_hits.reserve(anchor_ids_hits.size() + (e - b));
        for (uint64_t i = b; i < e; ++i) {
            uint32_t id = store.anchors[i];
            if (should_filter(filter, id)) continue;
            float final_score = hit.score * (f16_bits_to_f32(store.scores_f16[i]) / 100.0f);  // :426
            if (is_explain(options)) {  // :429-441
                std::vector<Explain>& vecco = res.explain[id];
                Explain e;
                e.kind = Explain::TermToAnchor;
                e.term_id = hit.id;
                e.a = hit.score;
                e.b = f16_bits_to_f32(store.scores_f16[i]) / 100.0f;
                e.c = final_score;
                vecco.push_back(e);
                auto exp = result.explain.find(hit.id);  // the dictionary result's records of this term
                if (exp != result.explain.end()) vecco.insert(vecco.end(), exp->second.begin(), exp->second.end());
            }
            anchor_ids_hits.push_back(Hit{id, final_score});
        }
    }
    // :453-464 sort by id, dedup keeping the max score
    std::stable_sort(anchor_ids_hits.begin(), anchor_ids_hits.end(), [](const Hit& a, const Hit& b) { return a.id < b.id; });
    {
        size_t w = 0;
        for (size_t r = 0; r < anchor_ids_hits.size(); ++r) {
            if (w > 0 && anchor_ids_hits[w - 1].id == anchor_ids_hits[r].id) {
                if (anchor_ids_hits[r].score > anchor_ids_hits[w - 1].score) anchor_ids_hits[w - 1].score = anchor_ids_hits[r].score;
            } else anchor_ids_hits[w++] = anchor_ids_hits[r];
        }
        anchor_ids_hits.resize(w);
    }
    // :468-498 ids only: text ids -> anchors
    std::vector<uint32_t> fast_field_res_ids;
    if (!result.hits_ids.empty()) {
        if (is_anchor_identity(index, options.path)) {
            fast_field_res_ids.insert(fast_field_res_ids.end(), result.hits_ids.begin(), result.hits_ids.end());
        } else {
            const KeyValueStore& t2a = index.get_valueid_to_parent(options.path + TEXT_ID_TO_ANCHOR);
            for (uint32_t id : result.hits_ids) {
                const uint32_t *b, *e;
                if (t2a.get_values(id, &b, &e)) fast_field_res_ids.insert(fast_field_res_ids.end(), b, e);
            }
        }
    }
    res.hits_ids = std::move(fast_field_res_ids);
    res.hits_scores = std::move(anchor_ids_hits);
    return res;
}

// =====================================================================================
// A3-A6 — set operations (src/search/set_op.rs)
// =====================================================================================
static TermIdHits merge_term_id_hits(std::vector<SearchFieldResult>& results) {  // :29-47
    TermIdHits out;
    for (auto& el : results) {
        for (auto& [attr, v] : el.term_id_hits_in_field) {
            auto& dst = out[attr];
            for (auto& [term, hits] : v) dst[term] = hits;
        }
        el.term_id_hits_in_field.clear();
    }
    return out;
}

static std::map<std::string, std::vector<std::string>> merge_term_id_texts(std::vector<SearchFieldResult>& results) {  // set_op.rs:49-63
    std::map<std::string, std::vector<std::string>> out;
    for (auto& el : results) {
        for (auto& kv : el.term_text_in_field) out[kv.first].insert(out[kv.first].end(), kv.second.begin(), kv.second.end());
        el.term_text_in_field.clear();
    }
    return out;
}
template <class V>
static size_t get_shortest_result(const std::vector<V>& lens) {  // :9-17 (first minimal)
    size_t idx = 0;
    uint64_t best = UINT64_MAX;
    for (size_t i = 0; i < lens.size(); ++i)
        if (uint64_t(lens[i]) < best) {
            best = uint64_t(lens[i]);
            idx = i;
        }
    return idx;
}
static void sort_hits_by_id(std::vector<Hit>& v) {
    if (!std::is_sorted(v.begin(), v.end(), [](const Hit& a, const Hit& b) { return a.id < b.id; }))
        std::stable_sort(v.begin(), v.end(), [](const Hit& a, const Hit& b) { return a.id < b.id; });
}

SearchFieldResult intersect_hits_score(std::vector<SearchFieldResult> and_results) {  // :368-446
    if (and_results.empty()) return SearchFieldResult{};
    if (and_results.size() == 1) return std::move(and_results[0]);
    const bool should_explain = is_explain(and_results[0].request);  // :384 (the first operand, before the shortest one is taken out)
    TermIdHits term_id_hits_in_field = merge_term_id_hits(and_results);
    auto term_text_in_field = merge_term_id_texts(and_results);
    std::vector<size_t> lens;
    for (auto& r : and_results) lens.push_back(r.hits_scores.size());
    size_t index_shortest = get_shortest_result(lens);
    for (auto& r : and_results) sort_hits_by_id(r.hits_scores);  // :390-392
    // swap_remove(index_shortest) :393
    std::vector<Hit> shortest_result = std::move(and_results[index_shortest].hits_scores);
    if (index_shortest != and_results.size() - 1) std::swap(and_results[index_shortest], and_results.back());
    and_results.pop_back();

    struct IterCur {
        const Hit* it;
        const Hit* end;
        Hit current;
    };
    std::vector<IterCur> iters;  // :399-408 (empty lists are dropped)
    for (auto& r : and_results) {
        if (r.hits_scores.empty()) continue;
        iters.push_back({r.hits_scores.data() + 1, r.hits_scores.data() + r.hits_scores.size(), r.hits_scores[0]});
    }
    auto check_score_iter_for_id = [](IterCur& ic, uint32_t current_id) -> bool {  // :347-366
        if (ic.current.id == current_id) return true;
        if (ic.current.id > current_id) return false;
        while (ic.it != ic.end) {
            Hit el = *ic.it++;
            ic.current = el;
            if (el.id > current_id) return false;
            if (el.id == current_id) return true;
        }
        return false;
    };
    std::vector<Hit> intersected_hits;
    intersected_hits.reserve(shortest_result.size());
    for (const Hit& cur : shortest_result) {  // :410-419
        bool all = true;
        for (auto& ic : iters)
            if (!check_score_iter_for_id(ic, cur.id)) {  // Iterator::all short-circuits
                all = false;
                break;
            }
        if (all) {
            float score = 0.0f;
            for (auto& ic : iters) score += ic.current.score;
            score += cur.score;
            intersected_hits.push_back(Hit{cur.id, score});
        }
    }
    SearchFieldResult res;
    if (should_explain) {  // :421-433: the records of the operands that are left after the shortest one was taken out
        for (const Hit& hit : intersected_hits)
            for (auto& r : and_results) {
                auto exp = r.explain.find(hit.id);
                if (exp != r.explain.end()) {
                    std::vector<Explain>& dst = res.explain[hit.id];
                    dst.insert(dst.end(), exp->second.begin(), exp->second.end());
                }
            }
    }
    res.term_id_hits_in_field = std::move(term_id_hits_in_field);
    res.term_text_in_field = std::move(term_text_in_field);
    res.hits_scores = std::move(intersected_hits);
    res.request = and_results[0].request;  // :439
    return res;
}

SearchFieldResult union_hits_score(std::vector<SearchFieldResult> or_results) {  // :87-220
    if (or_results.empty()) return SearchFieldResult{};
    if (or_results.size() == 1) return std::move(or_results[0]);
    TermIdHits term_id_hits_in_field = merge_term_id_hits(or_results);
    auto term_text_in_field = merge_term_id_texts(or_results);
    for (auto& r : or_results) sort_hits_by_id(r.hits_scores);  // :114-117
    std::vector<std::string> terms;  // :122-124
    for (auto& r : or_results) terms.push_back(r.request.terms.empty() ? std::string() : r.request.terms[0]);
    std::sort(terms.begin(), terms.end());
    terms.erase(std::unique(terms.begin(), terms.end()), terms.end());

    struct MiniHit {
        uint32_t id;
        float score;
        uint8_t term_id;
    };
    std::vector<std::vector<MiniHit>> lists(or_results.size());
    std::vector<std::pair<const MiniHit*, const MiniHit*>> sources;
    for (size_t i = 0; i < or_results.size(); ++i) {
        const std::string t = or_results[i].request.terms.empty() ? std::string() : or_results[i].request.terms[0];
        uint8_t term_id = uint8_t(std::find(terms.begin(), terms.end(), t) - terms.begin());  // :143
        lists[i].reserve(or_results[i].hits_scores.size());
        for (auto& h : or_results[i].hits_scores) lists[i].push_back({h.id, h.score, term_id});
        sources.push_back({lists[i].data(), lists[i].data() + lists[i].size()});
    }
    auto less = [](const MiniHit& a, const MiniHit& b) { return a.id < b.id; };
    KMerge<MiniHit, decltype(less)> mergo(sources, less);  // :159

    const bool should_explain = is_explain(or_results[0].request);  // :120
    ExplainMap explain_hits;
    if (should_explain)  // :133-137 HashMap::extend: an operand's entry REPLACES the entry an earlier operand had under the same key
        for (auto& r : or_results)
            for (auto& kv : r.explain) explain_hits[kv.first] = kv.second;
    std::vector<Hit> union_hits;
    std::vector<float> max_scores_per_term(terms.size(), 0.0f);
    MiniHit el;
    bool have = mergo.next(el);
    while (have) {  // group_by id :169-196
        uint32_t id = el.id;
        for (auto& m : max_scores_per_term) m = 0.0f;
        while (have && el.id == id) {
            float& m = max_scores_per_term[el.term_id];
            m = std::fmax(m, el.score);  // f32::max
            have = mergo.next(el);
        }
        float num_distinct_terms = 0.0f;
        for (float m : max_scores_per_term)
            if (m >= 0.00001f) num_distinct_terms += 1.0f;
        float sum = 0.0f;
        for (float m : max_scores_per_term) sum += m;
        float s = sum * num_distinct_terms * num_distinct_terms;  // :183
        union_hits.push_back(Hit{id, s});
        if (should_explain) {  // :187-195
            Explain e;
            e.kind = Explain::OrSumOverDistinctTerms;
            e.a = sum;
            explain_hits[id].push_back(e);
        }
    }
    if (should_explain)  // :199-208
        for (const Hit& hit : union_hits)
            for (auto& r : or_results) {
                auto exp = r.explain.find(hit.id);
                if (exp != r.explain.end()) {
                    std::vector<Explain>& dst = explain_hits[hit.id];
                    dst.insert(dst.end(), exp->second.begin(), exp->second.end());
                }
            }
    SearchFieldResult res;
    res.term_id_hits_in_field = std::move(term_id_hits_in_field);
    res.term_text_in_field = std::move(term_text_in_field);
    res.hits_scores = std::move(union_hits);
    res.explain = std::move(explain_hits);
    res.request = or_results[0].request;  // :215
    return res;
}

SearchFieldResult union_hits_ids(std::vector<SearchFieldResult> or_results) {  // :222-258
    if (or_results.empty()) return SearchFieldResult{};
    if (or_results.size() == 1) return std::move(or_results[0]);
    std::vector<std::pair<const uint32_t*, const uint32_t*>> sources;
    for (auto& r : or_results) {
        std::sort(r.hits_ids.begin(), r.hits_ids.end());
        sources.push_back({r.hits_ids.data(), r.hits_ids.data() + r.hits_ids.size()});
    }
    auto less = [](const uint32_t& a, const uint32_t& b) { return a < b; };
    KMerge<uint32_t, decltype(less)> mergo(sources, less);
    std::vector<uint32_t> union_hits;
    uint32_t id;
    while (mergo.next(id))
        if (union_hits.empty() || union_hits.back() != id) union_hits.push_back(id);
    SearchFieldResult res;
    res.hits_ids = std::move(union_hits);
    res.request = or_results[0].request;
    return res;
}

SearchFieldResult intersect_score_hits_with_ids(SearchFieldResult score_results, SearchFieldResult id_hits) {  // :311-326
    sort_hits_by_id(score_results.hits_scores);
    std::sort(id_hits.hits_ids.begin(), id_hits.hits_ids.end());
    if (!id_hits.hits_ids.empty()) {
        size_t pos = 1;
        const size_t n = id_hits.hits_ids.size();
        uint32_t current = id_hits.hits_ids[0];
        std::vector<Hit> kept;
        for (const Hit& hit : score_results.hits_scores) {
            while (current < hit.id) current = pos < n ? id_hits.hits_ids[pos++] : UINT32_MAX;  // unwrap_or(&u32::MAX)
            if (hit.id == current) kept.push_back(hit);
        }
        score_results.hits_scores = std::move(kept);
    }
    // NOTE :316: with an EMPTY id list the reference leaves the scored hits untouched (quirk, kept).
    return score_results;
}

SearchFieldResult intersect_hits_ids(std::vector<SearchFieldResult> and_results) {  // :468-509
    if (and_results.empty()) return SearchFieldResult{};
    if (and_results.size() == 1) return std::move(and_results[0]);
    std::vector<size_t> lens;
    for (auto& r : and_results) lens.push_back(r.hits_ids.size());
    size_t index_shortest = get_shortest_result(lens);
    for (auto& r : and_results) std::sort(r.hits_ids.begin(), r.hits_ids.end());
    std::vector<uint32_t> shortest_result = std::move(and_results[index_shortest].hits_ids);
    if (index_shortest != and_results.size() - 1) std::swap(and_results[index_shortest], and_results.back());
    and_results.pop_back();
    struct IterCur {
        const uint32_t* it;
        const uint32_t* end;
        uint32_t current;
    };
    std::vector<IterCur> iters;
    for (auto& r : and_results) {
        if (r.hits_ids.empty()) continue;
        iters.push_back({r.hits_ids.data() + 1, r.hits_ids.data() + r.hits_ids.size(), r.hits_ids[0]});
    }
    auto check = [](IterCur& ic, uint32_t current_id) -> bool {  // :448-466
        if (ic.current == current_id) return true;
        if (ic.current > current_id) return false;
        while (ic.it != ic.end) {
            uint32_t id = *ic.it++;
            ic.current = id;
            if (id > current_id) return false;
            if (id == current_id) return true;
        }
        return false;
    };
    std::vector<uint32_t> out;
    for (uint32_t id : shortest_result) {
        bool all = true;
        for (auto& ic : iters)
            if (!check(ic, id)) {
                all = false;
                break;
            }
        if (all) out.push_back(id);
    }
    SearchFieldResult res;  // :505-508 (request NOT carried)
    res.hits_ids = std::move(out);
    return res;
}

// =====================================================================================
// A7 — multiplicative boosts by id lists (src/search/boost.rs:89-237, 380-402)
// =====================================================================================
SearchFieldResult apply_boost_from_iter(SearchFieldResult results, const std::function<bool(Hit&)>& next) {  // :197-237
    const bool should_explain = is_explain(results.request);  // :200
    auto move_boost = [&](Hit& hit, Hit& hit_curr) {
        Hit b_hit;
        while (next(b_hit)) {
            if (b_hit.id > hit.id) {
                hit_curr = b_hit;
                break;
            } else if (b_hit.id == hit.id) {
                hit_curr = b_hit;
                hit.score *= b_hit.score;
                if (should_explain) push_boost_record(&results.explain, hit.id, b_hit.score);  // :213-217 (not for the entry the look-ahead rested on, :228)
            }
        }
    };
    Hit hit_curr;
    if (next(hit_curr)) {
        for (Hit& hit : results.hits_scores) {
            if (hit_curr.id < hit.id) {
                move_boost(hit, hit_curr);
            } else if (hit_curr.id == hit.id) {
                hit.score *= hit_curr.score;
                move_boost(hit, hit_curr);
            }
        }
    }
    return results;
}

SearchFieldResult boost_hits_ids_vec_multi(SearchFieldResult results, std::vector<SearchFieldResult>& boost) {  // :380-402
    sort_hits_by_id(results.hits_scores);
    std::vector<std::vector<Hit>> lists(boost.size());
    std::vector<std::pair<const Hit*, const Hit*>> sources;
    for (size_t i = 0; i < boost.size(); ++i) {
        auto& res = boost[i];
        sort_hits_by_id(res.hits_scores);
        std::sort(res.hits_ids.begin(), res.hits_ids.end());
        float boost_val = res.request.boost.value_or(2.0f);  // :393
        lists[i].reserve(res.hits_ids.size());
        for (uint32_t id : res.hits_ids) lists[i].push_back(Hit{id, boost_val});
        sources.push_back({lists[i].data(), lists[i].data() + lists[i].size()});
    }
    auto less = [](const Hit& a, const Hit& b) { return a.id < b.id; };
    KMerge<Hit, decltype(less)> mergo(sources, less);
    return apply_boost_from_iter(std::move(results), [&](Hit& h) { return mergo.next(h); });
}

SearchFieldResult apply_boost_term(const Index& index, SearchFieldResult res, const std::vector<RequestSearchPart>& boost_term) {  // :89-195
    // The LRU term_boost_cache (:92-171) only memoises `data`; results are identical without it.
    std::vector<SearchFieldResult> data;
    for (const RequestSearchPart& part : boost_term) {  // :174-187
        PlanRequestSearchPart req;
        req.request = part;
        req.get_ids = true;
        SearchFieldResult result = get_term_ids_in_field(index, req);
        result = resolve_token_to_anchor(index, req.request, nullptr, result);
        data.push_back(std::move(result));
    }
    return boost_hits_ids_vec_multi(std::move(res), data);
}

// =====================================================================================
// A8 — phrase pairs (src/search/search_field.rs:247-275)
// =====================================================================================
SearchFieldResult get_anchor_for_phrases_in_field(const Index& index, const std::string& path, const std::vector<uint32_t>& ids1,
                                                  const std::vector<uint32_t>& ids2) {
    SearchFieldResult result;
    const PhrasePairToAnchor& store = index.get_phrase_pair_to_anchor(path);
    for (uint32_t t1 : ids1)
        for (uint32_t t2 : ids2) {
            auto key = std::make_pair(t1, t2);
            auto it = std::lower_bound(store.keys.begin(), store.keys.end(), key);
            if (it != store.keys.end() && *it == key) {
                size_t k = size_t(it - store.keys.begin());
                result.hits_ids.insert(result.hits_ids.end(), store.anchors.begin() + store.offsets[k], store.anchors.begin() + store.offsets[k + 1]);
            }
        }
    std::sort(result.hits_ids.begin(), result.hits_ids.end());  // :273 (no dedup)
    return result;
}
static SearchFieldResult get_anchor_for_phrases_in_search_results(const Index& index, const std::string& path_in, const SearchFieldResult& res1,
                                                                  const SearchFieldResult& res2) {  // :247-261
    std::string path = path_in;
    if (!ends_with(path, TEXTINDEX)) path += TEXTINDEX;
    if (!ends_with(path, PHRASE_PAIR_TO_ANCHOR)) path += PHRASE_PAIR_TO_ANCHOR;
    return get_anchor_for_phrases_in_field(index, path, res1.hits_ids, res2.hits_ids);
}

// =====================================================================================
// A9 — text locality (src/search/boost.rs:11-87, src/search.rs:114-120)
// =====================================================================================
static std::vector<Hit> boost_text_locality(const Index& index, const std::string& path, std::map<std::string, std::vector<uint32_t>>& search_term_to_text_ids) {
    std::vector<Hit> boost_anchor;
    if (search_term_to_text_ids.size() <= 1) return boost_anchor;  // :36-39
    const KeyValueStore& token_to_text_id = index.get_valueid_to_parent(path + TOKENS_TO_TEXT_ID);
    std::vector<std::vector<uint32_t>> terms_text_ids;
    for (auto& [term, ids] : search_term_to_text_ids) {  // :45-49, get_all_value_ids search.rs:114-120
        std::vector<uint32_t> text_ids;
        for (uint32_t id : ids) {
            const uint32_t *b, *e;
            if (token_to_text_id.get_values(id, &b, &e)) text_ids.insert(text_ids.end(), b, e);
        }
        std::sort(text_ids.begin(), text_ids.end());
        terms_text_ids.push_back(std::move(text_ids));
    }
    std::vector<std::pair<const uint32_t*, const uint32_t*>> sources;
    for (auto& v : terms_text_ids) sources.push_back({v.data(), v.data() + v.size()});
    auto less = [](const uint32_t& a, const uint32_t& b) { return a < b; };
    KMerge<uint32_t, decltype(less)> mergo(sources, less);
    std::vector<std::pair<uint32_t, size_t>> boost_text_ids;
    uint32_t id;
    bool have = mergo.next(id);
    while (have) {  // :51-56
        uint32_t cur = id;
        size_t n = 0;
        while (have && id == cur) {
            ++n;
            have = mergo.next(id);
        }
        if (n > 1) boost_text_ids.push_back({cur, n});
    }
    std::stable_sort(boost_text_ids.begin(), boost_text_ids.end(), [](auto& a, auto& b) { return a.first < b.first; });
    if (is_anchor_identity(index, path)) {  // :60-71
        for (auto& t : boost_text_ids) boost_anchor.push_back(Hit{t.first, 2.0f * float(t.second) * float(t.second)});
    } else {  // :72-83
        const KeyValueStore& text_id_to_anchor = index.get_valueid_to_parent(path + TEXT_ID_TO_ANCHOR);
        for (auto& t : boost_text_ids) {
            const uint32_t *b, *e;
            if (text_id_to_anchor.get_values(t.first, &b, &e))
                for (const uint32_t* p = b; p != e; ++p) boost_anchor.push_back(Hit{*p, 2.0f * float(t.second) * float(t.second)});
        }
    }
    std::stable_sort(boost_anchor.begin(), boost_anchor.end(), [](const Hit& a, const Hit& b) { return a.id < b.id; });  // :85
    return boost_anchor;
}

std::vector<Hit> boost_text_locality_all(const Index& index, TermIdHits& term_id_hits_in_field) {  // :11-32
    std::vector<std::vector<Hit>> boosts;
    for (auto& [path, term_with_ids] : term_id_hits_in_field) boosts.push_back(boost_text_locality(index, path, term_with_ids));
    std::vector<std::pair<const Hit*, const Hit*>> sources;
    for (auto& v : boosts) sources.push_back({v.data(), v.data() + v.size()});
    auto less = [](const Hit& a, const Hit& b) { return a.id < b.id; };
    KMerge<Hit, decltype(less)> mergo(sources, less);
    std::vector<Hit> boost_anchor;
    Hit h;
    bool have = mergo.next(h);
    while (have) {
        uint32_t id = h.id;
        // :25 max_by with the comparator reversed (b.partial_cmp(a)) == the MINIMUM of the group
        float best = h.score;
        have = mergo.next(h);
        while (have && h.id == id) {
            if (h.score < best) best = h.score;
            have = mergo.next(h);
        }
        boost_anchor.push_back(Hit{id, best});
    }
    return boost_anchor;
}

// =====================================================================================
// A10 — boosts by indexed column (src/search/boost.rs:255-504, src/expression.rs)
// =====================================================================================
float score_expression(const std::string& expression, float rank) {  // expression.rs:26-95
    enum Op { Division, Mul, Add, Sub, Score, Float };
    struct Tok {
        Op op;
        float val;
    };
    std::vector<Tok> ops;
    std::string current;
    auto try_float = [&](const std::string& s, float& out) -> bool {
        if (s.empty()) return false;
        char* end = nullptr;
        out = std::strtof(s.c_str(), &end);
        return end && *end == 0 && end != s.c_str();
    };
    auto cps = vqtext::decode_utf8(expression);
    for (uint32_t next_char : cps) {
        if (next_char == ' ') {
            float v;
            if (try_float(current, v)) ops.push_back({Float, v});
            current.clear();
        }
        if (next_char != ' ') vqtext::append_utf8(current, next_char);
        if (current == "+") ops.push_back({Add, 0}), current.clear();
        else if (current == "-") ops.push_back({Sub, 0}), current.clear();
        else if (current == "/") ops.push_back({Division, 0}), current.clear();
        else if (current == "*") ops.push_back({Mul, 0}), current.clear();
        else if (current == "$SCORE") ops.push_back({Score, 0}), current.clear();
    }
    float v;
    if (try_float(current, v)) ops.push_back({Float, v});
    if (ops.size() < 3) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"bad score expression\" ");  // reference: index panic
    auto operand = [&](const Tok& t) -> float {
        if (t.op == Score) return rank;
        if (t.op == Float) return t.val;
        throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"Need to start with float oder $SCORE\" ");
    };
    float left = operand(ops[0]), right = operand(ops[2]);
    switch (ops[1].op) {
        case Division: return left / right;
        case Mul: return left * right;
        case Add: return left + right;
        case Sub: return left - right;
        default: throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"Need to be an operator [*, +, -, /]\" ");
    }
}

void apply_boost(Hit& hit, float boost_value, float boost_param, const std::optional<BoostFunction>& f, const std::optional<std::string>& expre,
                 ExplainMap* explain) {  // :283-377
    if (f) {
        switch (*f) {
            case BoostFunction::Log10:
                push_boost_record(explain, hit.id, std::log10(boost_value + boost_param));  // :297-300 (only this function records its factor)
                hit.score *= std::log10(boost_value + boost_param);
                break;
            case BoostFunction::Log2: hit.score *= std::log2(boost_value + boost_param); break;
            case BoostFunction::Multiply: hit.score *= boost_value + boost_param; break;
            case BoostFunction::Add: hit.score += boost_value + boost_param; break;
            case BoostFunction::Replace: hit.score = boost_value + boost_param; break;
        }
    }
    if (expre) hit.score += score_expression(*expre, boost_value);
    push_boost_record(explain, hit.id, hit.score);  // :371-374: the score after the boost
}

void apply_boost_values_anchor(SearchFieldResult& results, const RequestBoostPart& boost, const std::vector<Hit>& boost_values) {  // :255-281
    float boost_param = boost.param.value_or(0.0f);
    ExplainMap* explain = is_explain(results.request) ? &results.explain : nullptr;  // :258
    size_t pos = 0;
    auto next = [&](Hit& h) {
        if (pos >= boost_values.size()) return false;
        h = boost_values[pos++];
        return true;
    };
    Hit hit_curr;
    if (next(hit_curr)) {
        for (Hit& hit : results.hits_scores) {
            if (hit_curr.id < hit.id) {
                Hit b_hit;
                while (next(b_hit)) {
                    if (b_hit.id > hit.id) {
                        hit_curr = b_hit;
                        break;
                    } else if (b_hit.id == hit.id) {
                        hit_curr = b_hit;
                        apply_boost(hit, b_hit.score, boost_param, boost.boost_fun, boost.expression, explain);
                    }
                }
            } else if (hit_curr.id == hit.id) {
                apply_boost(hit, hit_curr.score, boost_param, boost.boost_fun, boost.expression, explain);
            }
        }
    }
}

void add_boost(const Index& index, const RequestBoostPart& boost, SearchFieldResult& hits) {  // :470-504
    const BoostStore& store = index.get_boost(boost.path + BOOST_VALID_TO_VALUE);
    float boost_param = boost.param.value_or(0.0f);
    std::vector<float> skip_when_score = boost.skip_when_score.value_or(std::vector<float>{});
    ExplainMap* explain = is_explain(hits.request) ? &hits.explain : nullptr;  // :484
    for (Hit& hit : hits.hits_scores) {
        bool skip = false;
        for (float x : skip_when_score)
            if (std::fabs(x - hit.score) < 0.00001f) {
                skip = true;
                break;
            }
        if (skip) continue;
        auto v = store.get_value(hit.id);
        if (v) {
            float boost_value;
            uint32_t bits = *v;
            std::memcpy(&boost_value, &bits, 4);
            apply_boost(hit, boost_value, boost_param, boost.boost_fun, boost.expression, explain);
        }
    }
}

// =====================================================================================
// A11 — top-n (src/search/sort.rs:5-22)
// =====================================================================================
std::vector<Hit> top_n_sort(std::vector<Hit> data, uint32_t top_n) {
    float worst_score = -std::numeric_limits<float>::max();
    std::vector<Hit> new_data;
    new_data.reserve(size_t(top_n) * 5 + 1);
    for (const Hit& el : data) {
        if (el.score < worst_score) continue;
        check_apply_top_n_sort(new_data, top_n, score_id_before, [&](const Hit& w) { worst_score = w.score; });
        new_data.push_back(el);
    }
    std::sort(new_data.begin(), new_data.end(), score_id_before);
    return new_data;
}
template <class T>
static void apply_top_skip(std::vector<T>& hits, std::optional<size_t> skip, std::optional<size_t> top) {  // search.rs:230-239
    if (skip) {
        size_t s = std::min(*skip, hits.size());
        hits.erase(hits.begin(), hits.begin() + s);
    }
    if (top) {
        size_t t = std::min(*top, hits.size());
        hits.resize(t);
    }
}

// =====================================================================================
// A12 / A14 — facets (src/facet.rs:14-93), path steps (src/util.rs:147-162)
// =====================================================================================
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

static std::string get_text_for_id(const Index& index, const std::string& path, uint32_t id) {  // search_field.rs:520-526
    auto it = index.fst.find(path);
    if (it == index.fst.end()) throw VelociError(ERR_FST_NOT_FOUND, "fst not found loaded in indices " + path + " ");
    if (id >= it->second.terms.size()) return std::string();
    return it->second.terms[id];
}

std::vector<std::pair<std::string, uint64_t>> get_facet(const Index& index, const FacetRequest& req, const std::vector<uint32_t>& ids) {
    std::vector<std::string> steps = get_steps_to_anchor(req.field);
    std::vector<std::pair<uint32_t, uint32_t>> groups;
    auto sort_and_apply_top_skip_group = [&](std::vector<std::pair<uint32_t, uint32_t>>& g) {  // :19-23
        // sort_unstable_by count desc: tie order is unspecified in the reference; value id asc here
        std::sort(g.begin(), g.end(), [](auto& a, auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
        apply_top_skip(g, std::nullopt, req.top);
    };
    if (steps.size() == 1 || index.has_index(steps.back() + ANCHOR_TO_TEXT_ID)) {  // :38-57
        std::string path = steps.size() == 1 ? steps.front() + PARENT_TO_VALUE_ID : steps.back() + ANCHOR_TO_TEXT_ID;
        const KeyValueStore& kv = index.get_valueid_to_parent(path);
        std::unordered_map<uint32_t, uint32_t> hits;  // count_values_for_ids persistence.rs:164-175
        for (uint32_t id : ids) {
            const uint32_t *b, *e;
            if (kv.get_values(id, &b, &e))
                for (const uint32_t* p = b; p != e; ++p) hits[*p] += 1;
        }
        groups.assign(hits.begin(), hits.end());
        sort_and_apply_top_skip_group(groups);
    } else {  // :59-70 join_anchor_to_leaf :75-93
        std::vector<uint32_t> level(ids.begin(), ids.end());
        for (size_t s = 0; s < steps.size(); ++s) {
            const KeyValueStore& kv = index.get_valueid_to_parent(steps[s] + PARENT_TO_VALUE_ID);
            std::vector<uint32_t> next;
            for (uint32_t id : level) {
                const uint32_t *b, *e;
                if (kv.get_values(id, &b, &e)) next.insert(next.end(), b, e);
            }
            level.swap(next);
        }
        std::sort(level.begin(), level.end());
        for (size_t i = 0; i < level.size();) {
            size_t j = i;
            while (j < level.size() && level[j] == level[i]) ++j;
            groups.push_back({level[i], uint32_t(j - i)});
            i = j;
        }
        sort_and_apply_top_skip_group(groups);
    }
    std::vector<std::pair<std::string, uint64_t>> out;  // :25-27
    for (auto& g : groups) out.push_back({get_text_for_id(index, steps.back(), g.first), g.second});
    return out;
}

// =====================================================================================
// index-time score (src/create/calculate_score.rs:34-49)
// =====================================================================================
uint32_t calculate_token_score_for_entry(uint32_t token_best_pos, uint32_t num_occurences, uint32_t num_tokens_in_text, bool is_exact) {
    float score = is_exact ? 400.0f : 2000.0f / (std::log2(float(token_best_pos) + 10.0f) + 10.0f);
    float num_occurence_modifier = std::log10(float(num_occurences) + 1000.0f) - 2.0f;
    num_occurence_modifier -= (num_occurence_modifier - 1.0f) * 0.7f;
    score /= num_occurence_modifier;
    float text_length_modifier = std::log10(float(num_tokens_in_text + 10));
    text_length_modifier -= (text_length_modifier - 1.0f) * 0.7f;
    score /= text_length_modifier;
    return uint32_t(score);
}

// =====================================================================================
// L3/L4 — plan semantics + search() (src/plan_creator/execution_plan.rs, plan_steps.rs, src/search.rs:143-228)
// =====================================================================================
namespace {

struct LeafEntry {
    PlanRequestSearchPart req;
    bool computed = false;
    SearchFieldResult result;
};

struct PlanCtx {
    const Index& index;
    const Request& request;
    std::map<std::string, LeafEntry> cache;  // FieldRequestCache, execution_plan.rs:13
    explicit PlanCtx(const Index& i, const Request& r) : index(i), request(r) {}

    void add_to_cache(const RequestSearchPart& part, bool ids_only) {  // :108-130
        auto it = cache.find(part.key());
        if (it != cache.end()) {
            it->second.req.get_ids |= ids_only;
            it->second.req.get_scores |= !ids_only;
            return;
        }
        LeafEntry e;
        e.req.request = part;
        e.req.get_scores = !ids_only;
        e.req.get_ids = ids_only;
        cache.emplace(part.key(), std::move(e));
    }
    void collect(const SearchRequest& r, bool ids_only) {  // :71-85
        if (r.kind == SearchRequest::Search) add_to_cache(r.part, ids_only);
        else
            for (auto& q : r.tree.queries) collect(q, ids_only);
    }
    LeafEntry& leaf(const RequestSearchPart& part) {
        auto it = cache.find(part.key());
        if (it == cache.end()) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"PlanCreator: Could not find request in field_search_cache\" ");
        return it->second;
    }
    // flag pass of plan_creator_search_part :401-418
    void flag_tree(const SearchRequest& r) {
        if (r.kind == SearchRequest::Search) {
            LeafEntry& e = leaf(r.part);
            e.req.store_term_id_hits |= (request.why_found || request.text_locality);
            e.req.store_term_texts |= request.why_found;  // :416
        } else
            for (auto& q : r.tree.queries) flag_tree(q);
    }
    const SearchFieldResult& field_result(const RequestSearchPart& part) {  // PlanStepFieldSearchToTokenIds, plan_steps.rs:137-148
        LeafEntry& e = leaf(part);
        if (!e.computed) {
            e.result = get_term_ids_in_field(index, e.req);
            e.computed = true;
        }
        return e.result;
    }

    // search_field.rs:640-689
    void resolve_token_hits_to_text_id_ids_only(const RequestSearchPart& options, SearchFieldResult& result) {
        std::string path = options.path;
        if (!ends_with(path, TEXTINDEX)) path += TEXTINDEX;
        auto cit = index.columns.find(extract_field_name(path));
        bool is_tokenized = cit != index.columns.end() && cit->second.tokenize;
        if (!is_tokenized) return;
        const KeyValueStore& token_kvdata = index.get_valueid_to_parent(path + TOKENS_TO_TEXT_ID);
        std::vector<uint32_t> token_hits;
        for (const Hit& hit : result.hits_scores) {
            const uint32_t *b, *e;
            if (token_kvdata.get_values(hit.id, &b, &e)) token_hits.insert(token_hits.end(), b, e);
            else token_hits.push_back(hit.id);
        }
        std::sort(token_hits.begin(), token_hits.end());
        token_hits.erase(std::unique(token_hits.begin(), token_hits.end()), token_hits.end());
        result.hits_ids = std::move(token_hits);
        result.hits_scores.clear();
    }
    // search.rs:281-315
    SearchFieldResult join_to_parent_ids(const SearchFieldResult& input, const std::string& path) {
        const KeyValueStore& kv = index.get_valueid_to_parent(path);
        std::vector<uint32_t> hits;
        for (uint32_t id : input.hits_ids) {
            const uint32_t *b, *e;
            if (kv.get_values(id, &b, &e)) hits.insert(hits.end(), b, e);
        }
        ExplainMap explain_hits;
        if (is_explain(input.request))  // search.rs:288-305: the value ids' records are looked up under the INPUT ids — a panic when one has none
            for (uint32_t id : input.hits_ids) {
                const uint32_t *b, *e;
                if (!kv.get_values(id, &b, &e)) continue;
                auto exp = input.explain.find(id);
                if (b != e && exp == input.explain.end())
                    throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"explain with a 1:n boost: could not find explain for id " + std::to_string(id) + " (the reference panics)\" ");
                for (const uint32_t* v = b; v != e; ++v) explain_hits.emplace(*v, exp->second);
            }
        std::sort(hits.begin(), hits.end());
        hits.erase(std::unique(hits.begin(), hits.end()), hits.end());
        SearchFieldResult res = SearchFieldResult::new_from(input);
        res.hits_ids = std::move(hits);
        res.explain = std::move(explain_hits);
        return res;
    }
    // boost.rs:432-468
    void get_boost_ids_and_resolve_to_anchor(const std::string& boost_path, SearchFieldResult& hits) {
        const BoostStore& store = index.get_boost(boost_path + BOOST_VALID_TO_VALUE);
        std::sort(hits.hits_ids.begin(), hits.hits_ids.end());
        for (uint32_t value_id : hits.hits_ids) {
            auto v = store.get_value(value_id);
            if (v) {
                float f;
                uint32_t bits = *v;
                std::memcpy(&f, &bits, 4);
                hits.boost_ids.push_back(Hit{value_id, f});
            }
        }
        hits.hits_ids.clear();
        std::vector<Hit> data;
        const KeyValueStore& kv = index.get_valueid_to_parent(boost_path + VALUE_ID_TO_ANCHOR);
        for (const Hit& bp : hits.boost_ids) {
            auto a = kv.get_value(bp.id);
            if (a) data.push_back(Hit{*a, bp.score});
        }
        hits.boost_ids = std::move(data);
    }

    // plan_creator_2 + plan_creator_search_part + the step implementations, executed depth-first
    SearchFieldResult exec(const SearchRequest& r, bool is_filter, const FilterResult* filter, std::vector<RequestBoostPart> boost) {
        if (r.kind != SearchRequest::Search) {
            std::vector<SearchFieldResult> results;
            for (auto& q : r.tree.queries) {
                std::vector<RequestBoostPart> child_boost = boost;  // merge_vec :263-270
                if (q.get_options() && q.get_options()->boost) child_boost.insert(child_boost.end(), q.get_options()->boost->begin(), q.get_options()->boost->end());
                results.push_back(exec(q, is_filter, filter, child_boost));
            }
            if (r.kind == SearchRequest::Or) return is_filter ? union_hits_ids(std::move(results)) : union_hits_score(std::move(results));
            return is_filter ? intersect_hits_ids(std::move(results)) : intersect_hits_score(std::move(results));
        }
        const RequestSearchPart& part = r.part;
        const SearchFieldResult& field_res = field_result(part);
        size_t pos = part.path.rfind("[]");
        if (pos != std::string::npos) {  // :422-509 1:n boost joined through the shared [] prefix
            std::string end_obj = part.path.substr(0, pos);
            std::vector<const RequestBoostPart*> boosto;
            for (auto& el : boost) {
                size_t p = el.path.rfind("[]");
                if (p != std::string::npos && el.path.substr(0, p) == end_obj) boosto.push_back(&el);
            }
            if (!boosto.empty()) {
                if (boosto.size() != 1) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"more than one boost matches the 1:n search path\" ");
                SearchFieldResult token_to_anchor = resolve_token_to_anchor(index, part, filter, field_res);
                // BoostToAnchor plan_steps.rs:174-197
                SearchFieldResult fr = field_res;
                resolve_token_hits_to_text_id_ids_only(part, fr);
                fr = join_to_parent_ids(fr, part.path + TEXTINDEX + VALUE_ID_TO_PARENT);
                get_boost_ids_and_resolve_to_anchor(boosto[0]->path, fr);
                // ApplyAnchorBoost plan_steps.rs:203-219
                apply_boost_values_anchor(token_to_anchor, *boosto[0], fr.boost_ids);
                return token_to_anchor;
            }
        }
        return resolve_token_to_anchor(index, part, filter, field_res);  // :511-533
    }
};

std::optional<std::string> highlight_document(const Index& index, const std::string& path, uint64_t value_id, const std::vector<uint32_t>& token_ids, const SnippetInfo& opt);

// get_why_found (src/search/why_found.rs:11-50): for every returned anchor and every searched field, the field's texts of that anchor
// (join_anchor_to_leaf, facet.rs:75-93) highlighted with ALL term ids the search matched in the field (highlight_document with the default snippet
// options); texts without a hit leave no entry
std::map<uint32_t, std::map<std::string, std::vector<std::string>>> get_why_found(const Index& index, const std::vector<uint32_t>& anchor_ids,
                                                                                  const TermIdHits& term_id_hits_in_field) {
    static const SnippetInfo kDefaultSnippetInfo;
    std::map<uint32_t, std::map<std::string, std::vector<std::string>>> anchor_highlights;
    for (auto& [path, term_with_ids] : term_id_hits_in_field) {
        const std::string field_name = extract_field_name(path);
        const std::vector<std::string> paths = get_steps_to_anchor(field_name);
        std::vector<uint32_t> all_term_ids_hits_in_path;
        for (auto& [term, hits] : term_with_ids) all_term_ids_hits_in_path.insert(all_term_ids_hits_in_path.end(), hits.begin(), hits.end());
        if (all_term_ids_hits_in_path.empty()) continue;
        for (uint32_t anchor_id : anchor_ids) {
            std::vector<uint32_t> ids(1, anchor_id);
            for (auto& step : paths) {
                const KeyValueStore& kv = index.get_valueid_to_parent(step + PARENT_TO_VALUE_ID);
                std::vector<uint32_t> next;
                for (uint32_t id : ids) {
                    const uint32_t *b, *e;
                    if (kv.get_values(id, &b, &e)) next.insert(next.end(), b, e);
                }
                ids.swap(next);
            }
            for (uint32_t value_id : ids) {
                auto highlighted_document = highlight_document(index, paths.back(), value_id, all_term_ids_hits_in_path, kDefaultSnippetInfo);
                if (highlighted_document) anchor_highlights[anchor_id][field_name].push_back(*highlighted_document);
            }
        }
    }
    return anchor_highlights;
}

}  // namespace

SearchResult search(Request request, const Index& index) {
    auto start = std::chrono::steady_clock::now();
    request.top = request.top ? request.top : std::optional<size_t>(10);  // :146
    if (!request.search_req) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"search_req is None, but is required in search\" ");

    if (request.explain) {  // execution_plan.rs:46-85: the header's flag goes into every part's options BEFORE the parts are collected (it is part of their equality)
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
        walk(*request.search_req);
        if (request.filter) walk(*request.filter);  // :98-103
    }
    PlanCtx ctx(index, request);
    // collect_all_field_request_into_cache :91-106
    if (request.phrase_boosts)
        for (auto& el : *request.phrase_boosts) {
            ctx.add_to_cache(el.search1, false);
            ctx.add_to_cache(el.search2, false);
        }
    ctx.collect(*request.search_req, false);
    if (request.filter) ctx.collect(*request.filter, true);
    // flags set while the plan is created (before anything executes)
    if (request.filter) ctx.flag_tree(*request.filter);
    ctx.flag_tree(*request.search_req);
    if (request.phrase_boosts)
        for (auto& el : *request.phrase_boosts) {  // :211-236
            ctx.leaf(el.search1).req.get_ids = true;
            ctx.leaf(el.search2).req.get_ids = true;
        }

    // filter subtree first :137-144
    std::optional<SearchFieldResult> filter_res;
    std::optional<FilterResult> filter;
    if (request.filter) {
        filter_res = ctx.exec(*request.filter, true, nullptr, {});
        filter = FilterResult::from_result(filter_res->hits_ids);  // send_result_to_channel plan_steps.rs:357-366
    }
    SearchFieldResult res = ctx.exec(*request.search_req, false, filter ? &*filter : nullptr, request.boost.value_or(std::vector<RequestBoostPart>{}));
    if (filter_res) res = intersect_score_hits_with_ids(std::move(res), *filter_res);  // :163-173
    if (request.boost) {  // :175-189 anchor-level boosts (paths without [])
        for (auto& b : *request.boost)
            if (b.path.find("[]") == std::string::npos) add_boost(index, b, res);
    }
    if (request.phrase_boosts) {  // :191-193, :202-262; plan_steps.rs:235-293
        std::vector<SearchFieldResult> boosts;
        for (auto& pb : *request.phrase_boosts) {
            const SearchFieldResult& r1 = ctx.field_result(pb.search1);
            const SearchFieldResult& r2 = ctx.field_result(pb.search2);
            if (pb.search1.path != pb.search2.path) throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"phrase boost over two different paths\" ");
            SearchFieldResult r = get_anchor_for_phrases_in_search_results(index, pb.search1.path, r1, r2);
            r.phrase_boost = std::make_pair(pb.search1.terms.empty() ? std::string() : pb.search1.terms[0],
                                            pb.search2.terms.empty() ? std::string() : pb.search2.terms[0]);
            boosts.push_back(std::move(r));
        }
        // sort_and_group_boosts_by_phrase_terms plan_steps.rs:235-258
        std::stable_sort(boosts.begin(), boosts.end(), [](const SearchFieldResult& a, const SearchFieldResult& b) { return *a.phrase_boost < *b.phrase_boost; });
        std::vector<SearchFieldResult> grouped;
        for (size_t i = 0; i < boosts.size();) {
            size_t j = i;
            std::vector<std::pair<const uint32_t*, const uint32_t*>> sources;
            while (j < boosts.size() && *boosts[j].phrase_boost == *boosts[i].phrase_boost) {
                sources.push_back({boosts[j].hits_ids.data(), boosts[j].hits_ids.data() + boosts[j].hits_ids.size()});
                ++j;
            }
            auto less = [](const uint32_t& a, const uint32_t& b) { return a < b; };
            KMerge<uint32_t, decltype(less)> mergo(sources, less);
            SearchFieldResult g;
            uint32_t id;
            while (mergo.next(id))
                if (g.hits_ids.empty() || g.hits_ids.back() != id) g.hits_ids.push_back(id);
            g.request.boost = 5.0f;  // plan_steps.rs:270-272
            grouped.push_back(std::move(g));
            i = j;
        }
        res = boost_hits_ids_vec_multi(std::move(res), grouped);
    }

    SearchResult search_result;
    search_result.explain = res.explain;  // search.rs:174 — before the term boosts and the text locality add their records
    if (request.boost_term) res = apply_boost_term(index, std::move(res), *request.boost_term);  // :176-178
    if (request.text_locality) {  // :180-184
        std::vector<Hit> boost_anchor = boost_text_locality_all(index, res.term_id_hits_in_field);
        size_t pos = 0;
        res = apply_boost_from_iter(std::move(res), [&](Hit& h) {
            if (pos >= boost_anchor.size()) return false;
            h = boost_anchor[pos++];
            return true;
        });
    }
    if (request.facets) {  // :188-206
        std::vector<uint32_t> hit_ids;
        hit_ids.reserve(res.hits_scores.size());
        for (auto& h : res.hits_scores) hit_ids.push_back(h.id);
        std::sort(hit_ids.begin(), hit_ids.end());
        search_result.has_facets = true;
        for (auto& fr : *request.facets) search_result.facets.push_back({fr.field, get_facet(index, fr, hit_ids)});
    }
    search_result.why_found_terms = res.term_text_in_field;  // :186
    search_result.num_hits = res.hits_scores.size();  // :207
    search_result.data = top_n_sort(std::move(res.hits_scores), uint32_t(*request.top) + uint32_t(request.skip.value_or(0)));  // :210-211
    apply_top_skip(search_result.data, request.skip, request.top);  // :218
    if (request.why_found && request.has_select) {  // :220-224
        std::vector<uint32_t> anchor_ids;
        for (auto& h : search_result.data) anchor_ids.push_back(h.id);
        search_result.why_found_info = get_why_found(index, anchor_ids, res.term_id_hits_in_field);
    }
    search_result.execution_time_ns = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - start).count());
    return search_result;
}

// search_field.rs:160-219: every part's matched terms (lower-cased texts, scores), texts merged keeping the best score, ranked by score
std::vector<SuggestEntry> suggest_multi(const Index& index, Request req) {
    if (!req.suggest) throw VelociError(ERR_INVALID_REQUEST, "only suggest allowed in suggest function");  // StringError, :196-198
    std::vector<SuggestEntry> out;
    for (auto& part : *req.suggest) {
        PlanRequestSearchPart p;
        p.request = part;
        p.get_scores = true;
        p.return_term = true;
        p.return_term_lowercase = true;
        SearchFieldResult r = get_term_ids_in_field(index, p);
        for (auto& h : r.hits_scores) out.push_back(SuggestEntry{r.terms.at(h.id), h.score, h.id});
    }
    // merge same text (:175-187): sort by text descending, a later duplicate hands its score to the kept one when larger
    std::stable_sort(out.begin(), out.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.text > b.text; });
    std::vector<SuggestEntry> merged;
    for (auto& e : out) {
        if (!merged.empty() && merged.back().text == e.text) {
            if (e.score > merged.back().score) merged.back().score = e.score;
        } else merged.push_back(e);
    }
    std::stable_sort(merged.begin(), merged.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.score > b.score; });  // :189 (unstable there)
    apply_top_skip(merged, req.skip, req.top);
    return merged;
}

// ------------------------------------------------------------------ highlight
namespace {
const SnippetInfo kDefaultSnippetInfo;  // search/request/snippet_info.rs:31-39

// highlight_field.rs:19-37
std::vector<std::vector<int64_t>> group_hit_positions_for_snippet(const std::vector<size_t>& hit_pos_of_tokens_in_doc, const SnippetInfo& opt) {
    const int64_t token_around_snippets = opt.num_words_around_snippet * 2;  // token separator token separator
    std::vector<std::vector<int64_t>> grouped;
    int64_t previous_token_pos = -token_around_snippets;
    for (size_t token_pos : hit_pos_of_tokens_in_doc) {
        if (int64_t(token_pos) - previous_token_pos >= token_around_snippets) grouped.emplace_back();
        previous_token_pos = int64_t(token_pos);
        grouped.back().push_back(int64_t(token_pos));
    }
    return grouped;
}
// highlight_field.rs:39-43
std::pair<size_t, size_t> grouped_to_positions_for_snippet(const std::vector<int64_t>& vec, size_t token_len, int64_t token_around_snippets) {
    const size_t start_index = size_t(std::max<int64_t>(vec.front() - token_around_snippets, 0));
    const size_t end_index = std::min(size_t(vec.back() + token_around_snippets + 1), token_len);
    return {start_index, end_index};
}

// highlight_field.rs:187-272: the snippet of text `value_id` around the tokens `token_ids`
std::optional<std::string> highlight_document(const Index& index, const std::string& path, uint64_t value_id, const std::vector<uint32_t>& token_ids, const SnippetInfo& opt) {
    const KeyValueStore& text_id_to_token_ids = index.get_valueid_to_parent(path + ".text_id_to_token_ids");
    const uint32_t *b, *e;
    if (!text_id_to_token_ids.get_values(value_id, &b, &e)) {  // :196-207
        if (std::find(token_ids.begin(), token_ids.end(), uint32_t(value_id)) != token_ids.end())
            return opt.snippet_start_tag + get_text_for_id(index, path, uint32_t(value_id)) + opt.snippet_end_tag;  // the whole text
        return std::nullopt;
    }
    const std::vector<uint32_t> documents_token_ids(b, e);
    const std::set<uint32_t> wanted(token_ids.begin(), token_ids.end());
    std::vector<size_t> hit_pos_of_tokens_in_doc;
    for (uint32_t token_id : wanted)  // :219-231
        for (size_t pos = 0; pos < documents_token_ids.size(); ++pos)
            if (documents_token_ids[pos] == token_id) hit_pos_of_tokens_in_doc.push_back(pos);
    if (hit_pos_of_tokens_in_doc.empty()) return std::nullopt;
    std::sort(hit_pos_of_tokens_in_doc.begin(), hit_pos_of_tokens_in_doc.end());

    const int64_t token_around_snippets = opt.num_words_around_snippet * 2;
    const auto grouped = group_hit_positions_for_snippet(hit_pos_of_tokens_in_doc, opt);
    // build_snippet (:45-71): the windows' texts, hits wrapped in the tags, at most max_snippets of them joined by the connector
    std::string snippet;
    size_t taken = 0;
    for (auto& g : grouped) {
        if (taken == opt.max_snippets) break;
        if (taken++) snippet += opt.snippet_connector;
        const auto window = grouped_to_positions_for_snippet(g, documents_token_ids.size(), token_around_snippets);
        for (size_t i = window.first; i < window.second; ++i) {
            const std::string text = get_text_for_id(index, path, documents_token_ids[i]);
            if (wanted.count(documents_token_ids[i])) snippet += opt.snippet_start_tag + text + opt.snippet_end_tag;
            else snippet += text;
        }
    }
    // ellipsis_snippet (:73-90)
    const int64_t first_index = int64_t(hit_pos_of_tokens_in_doc.front()), last_index = int64_t(hit_pos_of_tokens_in_doc.back());
    if (first_index > token_around_snippets) snippet.insert(0, opt.snippet_connector);
    if (last_index < int64_t(documents_token_ids.size()) - token_around_snippets) snippet += opt.snippet_connector;
    return snippet;
}
}  // namespace

// SimpleTokenizerGroupTokenIter (tokenizer/simple_tokenizer_group.rs:51-82) over DEFAULT_SEPERATORS (tokenizer/mod.rs:21-23): (token, is_separator)
static std::vector<std::pair<std::string, bool>> tokenize_group(const std::string& original) {
    static const uint32_t seps[] = {' ', '\t', '\n', '\r', ':', '(', ')', ',', '.', 0x2026, ';', 0x30FB, 0x2019, 0x2014, '-', '\\', '[', ']', '{', '}', '<', '>', '\'', '"', 0x201C, 0x2122};
    std::vector<std::pair<std::string, bool>> out;
    size_t last_returned_byte = 0;
    bool last_was_token = false;  // (the reference's name: true while inside a separator run)
    const std::vector<uint32_t> cps = vqtext::decode_utf8(original);
    size_t char_byte_pos = 0;
    for (uint32_t c : cps) {
        const bool is_sep = std::find(std::begin(seps), std::end(seps), c) != std::end(seps);
        if (is_sep) {
            if (char_byte_pos == 0) last_was_token = true;
            else if (!last_was_token) {
                out.push_back({original.substr(last_returned_byte, char_byte_pos - last_returned_byte), false});
                last_was_token = true;
                last_returned_byte = char_byte_pos;
            }
        } else if (last_was_token) {
            out.push_back({original.substr(last_returned_byte, char_byte_pos - last_returned_byte), true});
            last_was_token = false;
            last_returned_byte = char_byte_pos;
        }
        char_byte_pos += c < 0x80 ? 1 : c < 0x800 ? 2 : c < 0x10000 ? 3 : 4;
    }
    if (last_returned_byte != original.size()) out.push_back({original.substr(last_returned_byte), last_was_token});
    return out;
}

// highlight_field.rs:92-146
std::optional<std::string> highlight_text(const std::string& text, const std::set<std::string>& set, const SnippetInfo& opt, bool has_tokenizer) {
    if (opt.num_words_around_snippet < 0 || opt.num_words_around_snippet > 0x3FFFFFFF)
        throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"snippet_info.num_words_around_snippet out of range\" ");
    bool contains_any_token = false;
    if (set.size() == 1 && set.count(text)) return opt.snippet_start_tag + text + opt.snippet_end_tag;  // :96-98
    if (!has_tokenizer) return std::nullopt;                                                             // :99
    std::vector<std::string> tokens;
    std::vector<size_t> hit_pos_of_tokens_in_doc;
    for (auto& t : tokenize_group(text)) {
        if (set.count(t.first)) hit_pos_of_tokens_in_doc.push_back(tokens.size());
        tokens.push_back(t.first);
    }
    const int64_t token_around_snippets = opt.num_words_around_snippet * 2;
    const auto grouped = group_hit_positions_for_snippet(hit_pos_of_tokens_in_doc, opt);
    std::string snippet;
    size_t taken = 0;
    for (auto& g : grouped) {  // build_snippet :45-71
        if (taken == opt.max_snippets) break;
        if (taken++) snippet += opt.snippet_connector;
        const auto window = grouped_to_positions_for_snippet(g, tokens.size(), token_around_snippets);
        for (size_t i = window.first; i < window.second; ++i) {
            if (set.count(tokens[i])) {
                contains_any_token = true;
                snippet += opt.snippet_start_tag + tokens[i] + opt.snippet_end_tag;
            } else snippet += tokens[i];
        }
    }
    if (!hit_pos_of_tokens_in_doc.empty()) {  // ellipsis_snippet :73-90
        if (int64_t(hit_pos_of_tokens_in_doc.front()) > token_around_snippets) snippet.insert(0, opt.snippet_connector);
        if (int64_t(hit_pos_of_tokens_in_doc.back()) < int64_t(tokens.size()) - token_around_snippets) snippet += opt.snippet_connector;
    }
    if (contains_any_token) return snippet;
    return std::nullopt;
}

// search_field::highlight (search_field.rs:233-245) = get_term_ids_in_field + resolve_token_hits_to_text_id (:550-639) with snippets +
// get_text_score_id_from_result(false, ..) (:160-192)
std::vector<SuggestEntry> highlight(const Index& index, RequestSearchPart part) {
    for (auto& t : part.terms) t = vqtext::normalize_text(t);  // :234
    const SnippetInfo& opt = part.has_snippet_info ? part.snippet_info : kDefaultSnippetInfo;
    if (opt.num_words_around_snippet < 0 || opt.num_words_around_snippet > 0x3FFFFFFF)  // the reference's window arithmetic overflows / panics there
        throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"snippet_info.num_words_around_snippet out of range\" ");
    PlanRequestSearchPart options;
    options.request = part;
    options.get_scores = true;
    SearchFieldResult result = get_term_ids_in_field(index, options);
    std::map<uint32_t, std::string> highlight_of;  // SearchFieldResult::highlight

    // resolve_token_hits_to_text_id
    std::string path = options.request.path;
    if (!ends_with(path, TEXTINDEX)) path += TEXTINDEX;
    auto cit = index.columns.find(extract_field_name(path));
    const bool is_tokenized = cit != index.columns.end() && cit->second.tokenize;
    if (is_tokenized) {
        const bool add_snippets = options.request.snippet.value_or(false);
        const KeyValueStore& token_kvdata = index.get_valueid_to_parent(path + TOKENS_TO_TEXT_ID);
        struct TokenHit {
            uint32_t parent;
            float score;
            uint32_t token;
        };
        std::vector<TokenHit> token_hits;
        for (const Hit& hit : result.hits_scores) {
            const uint32_t *b, *e;
            if (token_kvdata.get_values(hit.id, &b, &e))
                for (const uint32_t* v = b; v != e; ++v) token_hits.push_back({*v, hit.score, hit.id});
        }
        std::stable_sort(token_hits.begin(), token_hits.end(), [](const TokenHit& a, const TokenHit& b) { return a.parent < b.parent; });  // :602 (unstable there)
        if (!token_hits.empty()) {
            if (add_snippets) result.hits_scores.clear();  // :608-610 only text hits for highlighting
            for (size_t i = 0; i < token_hits.size();) {
                size_t j = i;
                float max_score = token_hits[i].score;
                std::vector<uint32_t> tokens;
                for (; j < token_hits.size() && token_hits[j].parent == token_hits[i].parent; ++j) {
                    if (std::fabs(token_hits[j].score) >= std::fabs(max_score)) max_score = token_hits[j].score;  // max_by_key keeps the last maximum
                    tokens.push_back(token_hits[j].token);
                }
                result.hits_scores.push_back(Hit{token_hits[i].parent, max_score});
                if (add_snippets) {
                    auto doc = highlight_document(index, path, token_hits[i].parent, tokens, opt);
                    if (doc) highlight_of[token_hits[i].parent] = *doc;
                }
                i = j;
            }
        }
    }
    // get_text_score_id_from_result(false, ..): `&res.highlight[&id]` panics for a hit without a snippet
    std::vector<SuggestEntry> out;
    for (const Hit& h : result.hits_scores) {
        auto it = highlight_of.find(h.id);
        if (it == highlight_of.end())
            throw VelociError(ERR_INVALID_REQUEST, "InvalidRequest: \"highlight: hit " + std::to_string(h.id) + " has no snippet (the reference panics)\" ");
        out.push_back(SuggestEntry{it->second, h.score, h.id});
    }
    std::stable_sort(out.begin(), out.end(), [](const SuggestEntry& a, const SuggestEntry& b) { return a.score > b.score; });  // :189 (unstable there)
    apply_top_skip(out, options.request.skip, options.request.top);
    return out;
}

}  // namespace vo
