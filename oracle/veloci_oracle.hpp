// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// CPU restatement ("oracle") of veloci's query-execution path, single threaded, over plain
// decoded index arrays.  Every function cites the reference file:line it follows
// (paths relative to /root/reference, PSeitz/veloci @ 2024_10_08).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code; the product
// library (veloci_amd/csrc) never links, includes or calls anything in oracle/.
//
// Parity pinning: the reference is Rust and cannot be built here (no cargo/rustc), so this
// restatement is pinned against the known-answer vectors of the reference's own in-source unit
// tests (tests/golden/reference_unit_vectors.json).  Third-party arithmetic that is
// not under /root/reference is restated from the published algorithms and marked below:
//   * veloci_levenshtein_automata 0.1.0  -> Levenshtein / optimal-string-alignment distance
//   * itertools 0.12 kmerge_by           -> binary-heap k-way merge (tie order restated)
//   * half 2.3.1                         -> IEEE binary16 round-to-nearest-even
//   * fst 0.4.7                          -> sorted string table, ordinal == term id
//
// Shared with the product, and pinned on its own because of that: the request structs + JSON parser (request.hpp, json.hpp) and the Unicode
// lowercasing table (text.hpp) are included from veloci_amd/csrc — data-format plumbing, no query arithmetic.  A defect there would be
// invisible to product-vs-oracle parity, so tests/test_request_parse.py checks the parser against tests/golden/request_parse.json (what serde
// makes of 140+ request texts, derived by an independent restatement over Python's json module, tests/reqparse.py) and sweeps the lowercasing
// over every code point against Python's str.lower.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../veloci_amd/csrc/json.hpp"
#include "../veloci_amd/csrc/request.hpp"
#include "../veloci_amd/csrc/text.hpp"

namespace vo {

using namespace vqreq;  // Request structs, VelociError and error codes (wire-format plumbing shared with the product)

// ------------------------------------------------------------------ half (half 2.3.1: f16::from_f32 / to_f32)
inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) return uint16_t(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0));
    int32_t e = int32_t(exp) - 127 + 15;
    if (e >= 0x1F) return uint16_t(sign | 0x7C00u);  // overflow -> inf
    if (e <= 0) {
        if (e < -10) return uint16_t(sign);  // underflow -> 0
        man |= 0x800000u;
        uint32_t shift = uint32_t(14 - e);
        uint32_t half_man = man >> shift;
        uint32_t round_bit = 1u << (shift - 1);
        if ((man & round_bit) && ((man & (round_bit - 1)) || (half_man & 1))) half_man++;
        return uint16_t(sign | half_man);
    }
    uint32_t half = sign | (uint32_t(e) << 10) | (man >> 13);
    uint32_t round_bit = 0x1000u;
    if ((man & round_bit) && ((man & (round_bit - 1)) || (half & 1))) half++;  // RNE, may carry into exponent
    return uint16_t(half);
}
inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t(h) & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            man &= 0x3FFu;
            x = sign | (uint32_t(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 0x1F) x = sign | 0x7F800000u | (man << 13);
    else x = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// ------------------------------------------------------------------ core types
struct Hit {  // src/search.rs:53-57
    uint32_t id;
    float score;
};

using TermIdHits = std::map<std::string, std::map<std::string, std::vector<uint32_t>>>;  // path -> term -> term ids

struct Explain {  // src/search/result/explain.rs:2-21
    enum Kind { Boost, MaxTokenToTextId, TermToAnchor, LevenshteinScore, OrSumOverDistinctTerms } kind = Boost;
    float a = 0.0f, b = 0.0f, c = 0.0f;  // Boost(a) | TermToAnchor{term_score a, anchor_score b, final_score c} | LevenshteinScore{score a} | OrSum(a)
    uint32_t term_id = 0;
    std::string text;                    // LevenshteinScore.text_or_token_id
};
using ExplainMap = std::map<uint32_t, std::vector<Explain>>;  // FnvHashMap<u32, Vec<Explain>>: only looked up by key, never iterated for output
std::string explain_json(const std::vector<Explain>& records);  // serde's externally tagged form, floats as %.9g
inline bool is_explain(const RequestSearchPart& p) { return p.options && p.options->explain; }  // search_request.rs:182-184

struct SearchFieldResult {  // src/search/result/field_result.rs:7-30
    ExplainMap explain;
    std::vector<Hit> hits_scores;
    std::vector<uint32_t> hits_ids;
    std::vector<Hit> boost_ids;
    RequestSearchPart request;
    std::optional<std::pair<std::string, std::string>> phrase_boost;  // (search1.terms[0], search2.terms[0])
    TermIdHits term_id_hits_in_field;
    std::map<uint32_t, std::string> terms;                                 // term id -> text (return_term / store_term_texts, search_field.rs:347-353)
    std::map<std::string, std::vector<std::string>> term_text_in_field;    // path -> matched term texts (why_found, :386-389)
    static SearchFieldResult new_from(const SearchFieldResult& o) {  // :42-52
        SearchFieldResult r;
        r.explain = o.explain;  // :44 — keyed by whatever ids `o` had (term ids, when o is a dictionary result)
        r.request = o.request;
        r.phrase_boost = o.phrase_boost;
        r.term_id_hits_in_field = o.term_id_hits_in_field;
        r.terms = o.terms;
        r.term_text_in_field = o.term_text_in_field;
        return r;
    }
};

struct FilterResult {  // src/search/result/filter_result.rs:4-22
    bool is_set = false;
    std::unordered_set<uint32_t> set;
    std::vector<uint32_t> vec;
    static FilterResult from_result(const std::vector<uint32_t>& res) {
        FilterResult f;
        if (res.size() > 100000) {
            f.vec = res;
        } else {
            f.is_set = true;
            f.set.insert(res.begin(), res.end());
        }
        return f;
    }
};

struct SearchResult {  // src/search/result/search_result.rs:9-26
    uint64_t execution_time_ns = 0;
    uint64_t num_hits = 0;
    std::vector<Hit> data;
    bool has_facets = false;
    std::vector<std::pair<std::string, std::vector<std::pair<std::string, uint64_t>>>> facets;  // request order
    std::map<std::string, std::vector<std::string>> why_found_terms;  // search.rs:186 (the reference's map and list orders are unspecified)
    ExplainMap explain;                                               // search.rs:174; per hit: explain.get(&hit.id) (search.rs:86,96)
    // search.rs:220-224 (request.why_found && request.select): anchor -> field -> highlighted texts of the field that hold a matched token
    std::map<uint32_t, std::map<std::string, std::vector<std::string>>> why_found_info;
};
struct SuggestEntry {  // search_field.rs:158 SuggestFieldResult = Vec<(String, Score, TermId)>
    std::string text;
    float score;
    uint32_t term_id;
};
std::optional<std::string> highlight_text(const std::string& text, const std::set<std::string>& set, const SnippetInfo& opt, bool has_tokenizer);  // highlight_field.rs:92-146
std::vector<SuggestEntry> highlight(const struct Index& index, RequestSearchPart part);  // search_field.rs:233-245, highlight_field.rs:187-272
std::vector<SuggestEntry> suggest_multi(const struct Index& index, Request req);  // search_field.rs:194-219 (and :221-231 for a bare RequestSearchPart)

// ------------------------------------------------------------------ decoded index (Persistence, src/persistence.rs:52-72)
struct Fst {  // sorted term table standing in for fst::Map (ordinal == term id)
    std::vector<std::string> terms;
    // lowercase(term) -> ascending term ids; built once by finalize().  Stands in for the FST walk of an
    // exact, case-insensitive lookup (O(len) in the reference) so the CPU baseline is not charged a scan.
    std::unordered_map<std::string, std::vector<uint32_t>> lower_map;
    void finalize() {
        lower_map.clear();
        lower_map.reserve(terms.size());
        for (uint32_t i = 0; i < terms.size(); ++i) lower_map[vqtext::to_lower_utf8(terms[i])].push_back(i);
    }
};
struct TokenToAnchorScore {  // src/indices/persistence_score/token_to_anchor_score_vint.rs
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> anchors;
    std::vector<uint16_t> scores_f16;  // f16::from_f32(score as f32) :155
};
struct KeyValueStore {  // IndexIdToParent<Output=u32>, src/persistence.rs:142-181
    uint32_t key_base = 0;
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> values;
    // get_values: None when out of range or empty bucket (src/indices/indirect/indirect.rs:72-89)
    bool get_values(uint64_t id, const uint32_t** b, const uint32_t** e) const {
        if (id < key_base) return false;
        uint64_t row = id - key_base;
        if (row + 1 >= offsets.size()) return false;
        if (offsets[row] == offsets[row + 1]) return false;
        *b = values.data() + offsets[row];
        *e = values.data() + offsets[row + 1];
        return true;
    }
    std::optional<uint32_t> get_value(uint64_t id) const {  // src/persistence.rs:177-180
        const uint32_t *b, *e;
        if (!get_values(id, &b, &e)) return std::nullopt;
        return *b;
    }
};
struct PhrasePairToAnchor {  // src/indices/persistence_data_binary_search.rs:167-202
    std::vector<std::pair<uint32_t, uint32_t>> keys;  // sorted
    std::vector<uint64_t> offsets;
    std::vector<uint32_t> anchors;
};
struct BoostStore {  // boost_valueid_to_value
    uint32_t key_base = 0;
    std::vector<uint8_t> present;
    std::vector<uint32_t> bits;
    std::optional<uint32_t> get_value(uint64_t id) const {
        if (id < key_base) return std::nullopt;
        uint64_t row = id - key_base;
        if (row >= bits.size()) return std::nullopt;
        if (!present.empty() && !present[row]) return std::nullopt;
        return bits[row];
    }
};
struct ColumnMeta {
    bool is_anchor_identity_column = false;
    bool tokenize = true;
};

struct Index {
    uint32_t num_anchors = 0;
    std::map<std::string, Fst> fst;
    std::map<std::string, TokenToAnchorScore> token_to_anchor_score;
    std::map<std::string, KeyValueStore> key_value_stores;
    std::map<std::string, PhrasePairToAnchor> phrase_pair_to_anchor;
    std::map<std::string, BoostStore> boost_valueid_to_value;
    std::map<std::string, ColumnMeta> columns;

    bool has_index(const std::string& path) const { return key_value_stores.count(path) != 0; }  // src/persistence.rs has_index
    const KeyValueStore& get_valueid_to_parent(const std::string& path) const {  // src/persistence.rs:450-459
        auto it = key_value_stores.find(path);
        if (it == key_value_stores.end()) throw VelociError(ERR_INDEX_NOT_FOUND, "Did not found path in indices " + path);
        return it->second;
    }
    const TokenToAnchorScore& get_token_to_anchor(const std::string& path) const;
    const PhrasePairToAnchor& get_phrase_pair_to_anchor(const std::string& path) const;
    const BoostStore& get_boost(const std::string& path) const;
};

// suffix constants, src/persistence.rs:23-50
static const char* const TOKENS_TO_TEXT_ID = ".tokens_to_text_id";
static const char* const TEXT_ID_TO_TOKEN_IDS = ".text_id_to_token_ids";
static const char* const TO_ANCHOR_ID_SCORE = ".to_anchor_id_score";
static const char* const PHRASE_PAIR_TO_ANCHOR = ".phrase_pair_to_anchor";
static const char* const VALUE_ID_TO_PARENT = ".value_id_to_parent";
static const char* const PARENT_TO_VALUE_ID = ".parent_to_value_id";
static const char* const TEXT_ID_TO_ANCHOR = ".text_id_to_anchor";
static const char* const ANCHOR_TO_TEXT_ID = ".anchor_to_text_id";
static const char* const BOOST_VALID_TO_VALUE = ".boost_valid_to_value";
static const char* const VALUE_ID_TO_ANCHOR = ".value_id_to_anchor";
static const char* const TOKEN_VALUES = ".token_values";
static const char* const TEXTINDEX = ".textindex";

// ------------------------------------------------------------------ the restated functions
// A1
float get_default_score_for_distance(uint8_t distance, bool prefix_matches);       // search_field.rs:27-33
uint8_t distance(const std::string& s1, const std::string& s2);                     // search_field.rs:705-732
uint32_t levenshtein_cps(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b, bool transposition, bool ci);
struct PlanRequestSearchPart {  // plan_creator/execution_plan.rs:16-44
    RequestSearchPart request;
    bool get_scores = false, get_ids = false, store_term_id_hits = false;
    bool return_term = false, return_term_lowercase = false, store_term_texts = false;
};
SearchFieldResult get_term_ids_in_field(const Index&, PlanRequestSearchPart& options);  // search_field.rs:277-398
// A2
SearchFieldResult resolve_token_to_anchor(const Index&, const RequestSearchPart& options, const FilterResult* filter,
                                          const SearchFieldResult& result);  // search_field.rs:400-504
// A3-A6
SearchFieldResult intersect_hits_score(std::vector<SearchFieldResult> and_results);            // set_op.rs:368-446
SearchFieldResult union_hits_score(std::vector<SearchFieldResult> or_results);                 // set_op.rs:87-220
SearchFieldResult intersect_hits_ids(std::vector<SearchFieldResult> and_results);              // set_op.rs:468-509
SearchFieldResult union_hits_ids(std::vector<SearchFieldResult> or_results);                   // set_op.rs:222-258
SearchFieldResult intersect_score_hits_with_ids(SearchFieldResult score_results, SearchFieldResult id_hits);  // set_op.rs:311-326
// A7
SearchFieldResult apply_boost_from_iter(SearchFieldResult results, const std::function<bool(Hit&)>& next);  // boost.rs:197-237
SearchFieldResult boost_hits_ids_vec_multi(SearchFieldResult results, std::vector<SearchFieldResult>& boost);  // boost.rs:380-402
SearchFieldResult apply_boost_term(const Index&, SearchFieldResult res, const std::vector<RequestSearchPart>& boost_term);  // boost.rs:89-195
// A8
SearchFieldResult get_anchor_for_phrases_in_field(const Index&, const std::string& path, const std::vector<uint32_t>& ids1,
                                                  const std::vector<uint32_t>& ids2);  // search_field.rs:263-275
// A9
std::vector<Hit> boost_text_locality_all(const Index&, TermIdHits& term_id_hits_in_field);  // boost.rs:11-32
// A10
float score_expression(const std::string& expression, float rank);  // expression.rs:26-95
void apply_boost(Hit& hit, float boost_value, float boost_param, const std::optional<BoostFunction>& f,
                 const std::optional<std::string>& expre, ExplainMap* explain = nullptr);  // boost.rs:283-377
void apply_boost_values_anchor(SearchFieldResult& results, const RequestBoostPart& boost, const std::vector<Hit>& boost_values);  // boost.rs:255-281
void add_boost(const Index&, const RequestBoostPart& boost, SearchFieldResult& hits);  // boost.rs:470-504
// A11
std::vector<Hit> top_n_sort(std::vector<Hit> data, uint32_t top_n);  // sort.rs:5-22
// A12
std::vector<std::pair<std::string, uint64_t>> get_facet(const Index&, const FacetRequest& req, const std::vector<uint32_t>& ids);  // facet.rs:31-73
// A14
std::vector<std::string> get_steps_to_anchor(const std::string& path);  // util.rs:147-162
// index-time score, used by the synthetic generator's fixtures (create/calculate_score.rs:34-49)
uint32_t calculate_token_score_for_entry(uint32_t token_best_pos, uint32_t num_occurences, uint32_t num_tokens_in_text, bool is_exact);
// L4
SearchResult search(Request request, const Index& index);  // search.rs:143-228

}  // namespace vo
