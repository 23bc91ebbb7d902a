"""Generates tests/golden/reference_query_generator.json: the requests `query_generator::search_query` produces for the reference's own
query-generator tests (tests/all/test_query_generator.rs, tests/all/test_code_search.rs:73-150, tests/all/test_scores.rs:157-237), derived with the restatement in
tests/qgen.py, next to what each test asserts.  Run from the repo root: python tests/gen_query_generator_fixtures.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import qgen  # noqa: E402

QG_DOCS = [  # tests/all/test_query_generator.rs:47-137
    {"commonness": 123456, "ent_seq": "99999", "tags": ["nice", "cool"]},
    {"ent_seq": "1337", "commonness": 20, "tags": ["nice", "cool", "ent_seq:99999"],
     "kanji": [{"text": "偉容", "commonness": 0}, {"text": "威容", "commonness": 5}],
     "kana": [{"text": "いよう", "romaji": "Iyou", "commonness": 5}],
     "meanings": {"eng": ["will testo"], "ger": ["majestätischer Anblick (m)", "majestätisches Aussehen (n)", "Majestät (f)"]}},
    {"ent_seq": "1587690", "commonness": 20, "tags": ["nice"],
     "kanji": [{"text": "意欲", "commonness": 40}, {"text": "意慾", "commonness": 0}],
     "kana": [{"text": "いよく", "romaji": "Iyoku", "commonness": 40}],
     "meanings": {"eng": ["will", "urge", "having a long torso"], "ger": ["Wollen (n)", "Wille (m)", "Begeisterung (f)", "begeistern"]}},
    {"id": 1234566, "tags": ["awesome", "cool"], "commonness": 500, "kanji": [{"text": "意慾", "commonness": 20}], "kana": [{"text": "いよく"}], "ent_seq": "1587700"},
    {"commonness": 515151, "ent_seq": "25", "tags": ["nice", "cool"]},
    {"commonness": 30, "title": "COllectif", "meanings": {"ger": ["boostemich"]}},
    {"commonness": 30, "float_value": 5.123, "ent_seq": "26", "tags": ["nice", "coolo"]},
    {"commonness": 20, "ent_seq": "27", "my_bool": True, "tags": ["Eis", "cool"]},
    {"commonness": 20, "ent_seq": "28", "tags": ["nice", "cool"]},
]
QG_INDICES = {  # :11-39 (TOML there)
    "*GLOBAL*": {"features": ["All"]}, "commonness": {"facet": True, "boost": {"boost_type": "f32"}}, "ent_seq": {"fulltext": {"tokenize": True}},
    "nofulltext": {"fulltext": {"tokenize": False}}, "tags[]": {"facet": True}, "field1[].rank": {"boost": {"boost_type": "f32"}}, "field1[].text": {"tokenize": True},
    "kanji[].text": {"tokenize": True}, "meanings.ger[]": {"stopwords": ["stopword"], "fulltext": {"tokenize": True}}, "meanings.eng[]": {"fulltext": {"tokenize": True}},
    "kanji[].commonness": {"boost": {"boost_type": "f32"}}, "kana[].commonness": {"boost": {"boost_type": "f32"}},
}
CODE_DOCS = [{"line_number": 1, "line": "function myfun(param1: Type1)", "filename": "cool.ts", "filepath": "all/the/path"}]  # test_code_search.rs:32-41
CODE_INDICES = {"*GLOBAL*": {"features": ["All"]}, "filepath": {"fulltext": {"tokenize": True, "tokenize_on_chars": ["/", "\\"]}}, "filename": {"fulltext": {"tokenize": True}},
                "line": {"fulltext": {"tokenize": True}}, "line_number": {"boost": {"boost_type": "f32"}}}

SCORE_DOCS = [  # tests/all/test_scores.rs:6-45
    {"id": 1, "order": 500, "title": "greg tagebuch 05"},
    {"id": 2, "order": 20, "title": "and some some text 05 this is not relevant let tagebuch greg"},
    {"id": 3, "order": 1000, "title": "greg tagebuch"},
    {"id": 4, "commonness": 41, "meanings": {"ger": [{"text": "Fernsehen-Schauen (n)", "boost": 20}]}},
    {"id": 5, "commonness": 551, "meanings": {"ger": ["welch"]}},
    {"id": 6, "commonness": 2, "meanings": {"ger": ["weich"]}},
]
SCORE_INDICES = {  # :50-61 (TOML there)
    "title": {"fulltext": {"tokenize": True}}, "meanings.ger[].boost": {"boost": {"boost_type": "f32"}}, "meanings.ger[].text": {"fulltext": {"tokenize": True}},
    "commonness": {"boost": {"boost_type": "f32"}}, "order": {"boost": {"boost_type": "f32"}},
}

E = lambda n, d=None, **kw: dict({"len": n}, **({"doc": d} if d else {}), **kw)
URGE_DOC = [[0, ["ent_seq"], "1587690"], [0, ["commonness"], 20], [0, ["tags"], ["nice"]]]
T = "tests/all/test_query_generator.rs"
CASES = [
    ("simple_search_querygenerator_explained", "test_querygenerator", T + ":139-152", {"explain": True, "search_term": "urge"}, E(1, URGE_DOC, explain_len0=5)),
    ("simple_search_querygenerator_or_connect_explained", "test_querygenerator", T + ":154-168", {"explain": True, "search_term": "urge OR いよく"}, E(3, URGE_DOC, explain_len0=7)),
    ("simple_search_querygenerator", "test_querygenerator", T + ":169-179", {"search_term": "urge"}, E(1, URGE_DOC)),
    ("attributed_search", "test_querygenerator", T + ":181-190", {"search_term": "ent_seq:99999"}, E(1, [[0, ["ent_seq"], "99999"]])),
    ("disabled_attributed_search", "test_querygenerator", T + ":191-205", {"search_term": "ent_seq:99999", "parser_options": {"no_attributes": True}}, E(1, [[0, ["ent_seq"], "1337"]])),
    ("simple_search_querygenerator_or_connect", "test_querygenerator", T + ":207-217", {"search_term": "urge OR いよく"}, E(3, URGE_DOC)),
    ("simple_search_querygenerator_and", "test_querygenerator", T + ":219-229", {"search_term": "urge AND いよく"}, E(1, URGE_DOC)),
    ("simple_search_querygenerator_and_emtpy_stopword_list", "test_querygenerator", T + ":230-241", {"stopword_lists": [], "search_term": "urge AND いよく"}, E(1, URGE_DOC)),
    ("simple_search_querygenerator_and_stopword_list", "test_querygenerator", T + ":242-253", {"stopword_lists": ["en"], "search_term": "urge AND いよく"}, E(1, URGE_DOC)),
    ("simple_search_querygenerator_and_stopword_list_from_json", "test_querygenerator", T + ":255-269", {"stopword_lists": ["en"], "search_term": "urge AND いよく"}, E(1, URGE_DOC)),
    ("complex_search_querygenerator_from_json_1", "test_querygenerator", T + ":271-284",
     {"search_term": "will", "top": 10, "facets": ["commonness", "kanji[].commonness"], "levenshtein": 0, "boost_fields": {"meanings.eng[]": 1.5}}, E(2, [[0, ["meanings", "eng", 0], "will"]])),
    ("complex_search_querygenerator_from_json_2", "test_querygenerator", T + ":286-299",
     {"search_term": "will", "top": 10, "facets": ["commonness", "kanji[].commonness"], "levenshtein": 0, "boost_fields": {"meanings.eng[]": 1.5},
      "boost_terms": {"meanings.ger[]:majestätisches Aussehen (n)": 20.0}}, E(2, [[0, ["meanings", "eng", 0], "will testo"]])),
    ("simple_search_querygenerator_and_no_hit", "test_querygenerator", T + ":301-308", {"search_term": "urge AND いよく AND awesome"}, E(0)),
    ("simple_search_wildcard_starts_with_1", "test_querygenerator", T + ":310-316", {"search_term": "awes*"}, E(1)),
    ("simple_search_wildcard_starts_with_2", "test_querygenerator", T + ":318-320", {"search_term": "いよ*"}, E(3)),
    ("simple_search_wildcard_starts_with_with_levenshtein", "test_querygenerator", T + ":323-330", {"search_term": "awesam*"}, E(1)),
    ("contains_search_with_regex_starts_with", "test_querygenerator", T + ":332-340", {"search_term": "*wesom*", "fields": ["tags[]"]}, E(1)),
    ("contains_search_with_regex", "test_querygenerator", T + ":342-350", {"search_term": "*we*some", "fields": ["tags[]"]}, E(1)),
    ("contains_search_has_no_levenshtein", "test_querygenerator", T + ":352-360", {"search_term": "tags[]:*wesam*"}, E(0)),
    ("no_matching_fields_from_field_list", "test_querygenerator", T + ":362-371", {"search_term": "awes*", "fields": ["notexistingfield"]}, {"generator_error_contains": "All fields filtered"}),
    ("no_matching_fields_from_query", "test_querygenerator", T + ":373-381", {"search_term": "notexistingfield:awes*"}, {"generator_error_contains": "Field notexistingfield not found in"}),
    ("pattern_code_search_query_generator", "codeTest", "tests/all/test_code_search.rs:73-81", {"search_term": "*myfun*Type1*"}, E(1, [[0, ["line"], "function myfun(param1: Type1)"]])),
    ("pattern_code_search_ignore_case_query_generator", "codeTest", "tests/all/test_code_search.rs:83-91", {"search_term": "*myfun*type1*"}, E(1, [[0, ["line"], "function myfun(param1: Type1)"]])),
    ("pattern_code_search_case_sensitive_query_generator", "codeTest", "tests/all/test_code_search.rs:93-103", {"search_term": "*myfun*type1*", "ignore_case": False}, E(0)),
    ("pattern_code_search_no_fuzzy_query_generator", "codeTest", "tests/all/test_code_search.rs:105-112", {"search_term": "*myfun*type2*"}, E(0)),
    ("token_code_search_query_generator", "codeTest", "tests/all/test_code_search.rs:114-121", {"search_term": "myfun"}, E(1)),
    ("token_code_search_disable_parser_query_generator", "codeTest", "tests/all/test_code_search.rs:123-138",
     {"search_term": "*myfun(param1: Type1)*", "parser_options": {"no_parentheses": True, "no_attributes": True, "no_levensthein": True}}, E(1)),
    ("token_code_phrase_pattern_query_generator", "codeTest", "tests/all/test_code_search.rs:140-148", {"search_term": "\"*myfun(param1: Type1)*\""}, E(1)),
    # the reference's arithmetic pins: the scores themselves
    ("check_score_boost_relative_field", "test_score", "tests/all/test_scores.rs:157-183",
     {"search_term": "schauen", "fields": ["meanings.ger[].text"], "top": 3, "skip": 0, "why_found": True,
      "boost_queries": [{"path": "meanings.ger[].boost", "boost_fun": "Log10", "param": 10}], "boost_fields": {"meanings.ger[].text": 2.0}}, {"score0_gt": 40.0}),
    ("check_score_boost_add_value_from_field", "test_score", "tests/all/test_scores.rs:185-211",
     {"search_term": "weich", "fields": ["meanings.ger[]"], "levenshtein": 0, "boost_queries": [{"path": "commonness", "boost_fun": "Add"}]},
     {"score0_eq_base": {"base_params": {"search_term": "weich", "levenshtein": 0, "fields": ["meanings.ger[]"]}, "op": "add", "value": 2.0}}),
    ("check_score_boost_multiply_value_from_field", "test_score", "tests/all/test_scores.rs:213-237",
     {"search_term": "weich", "fields": ["meanings.ger[]"], "levenshtein": 0, "boost_queries": [{"path": "commonness", "boost_fun": "Multiply"}]},
     {"score0_eq_base": {"base_params": {"search_term": "weich", "levenshtein": 0, "fields": ["meanings.ger[]"]}, "op": "mul", "value": 2.0}}),
]


def fields_of(docs, indices, token_values=None):
    from veloci_amd import mini_indexer
    data, info = mini_indexer.build_index(docs, indices, token_values=token_values)
    all_fields = sorted(info.keys())
    search_fields = [f for f in all_fields if (f + ".textindex.to_anchor_id_score") in data.token_to_anchor_score]
    return all_fields, search_fields


def main():
    corpora = {
        "test_querygenerator": {"source": T + ":9-137", "indices": QG_INDICES, "docs": QG_DOCS, "token_values": [[{"text": "Begeisterung", "value": 20}], "meanings.ger[]"]},
        "codeTest": {"source": "tests/all/test_code_search.rs:11-41", "indices": CODE_INDICES, "docs": CODE_DOCS},
        "test_score": {"source": "tests/all/test_scores.rs:6-64", "indices": SCORE_INDICES, "docs": SCORE_DOCS},
    }
    fields = {name: fields_of(c["docs"], c["indices"], tuple(c["token_values"]) if c.get("token_values") else None) for name, c in corpora.items()}
    for name, c in corpora.items():
        c["all_fields"], c["search_fields"] = fields[name]
    cases = []
    for name, corpus, source, params, expect in CASES:
        case = {"name": name, "corpus": corpus, "source": source, "params": params, "expect": expect}
        try:
            case["request"] = qgen.search_query(*fields[corpus], params)
            if "score0_eq_base" in expect:  # the unboosted request of the same test, from its own parameters
                expect = dict(expect, score0_eq_base=dict(expect["score0_eq_base"], request=qgen.search_query(*fields[corpus], expect["score0_eq_base"]["base_params"])))
                case["expect"] = expect
        except qgen.GeneratorError as e:
            case["generator_error"] = str(e)
        cases.append(case)
    out = {"_about": "The reference's query-generator tests as data: `params` are the SearchQueryGeneratorParameters of each test, `request` is the search::Request "
                     "query_generator::search_query builds from them (src/query_generator.rs:175-246, query_generator/query_parser_to_veloci_request.rs:11-109; derived "
                     "by tests/gen_query_generator_fixtures.py with the restatement tests/qgen.py; the field order inside an expansion is unspecified in the reference "
                     "— FnvHashMap iteration — and sorted here), `expect` is what the test asserts.",
           "corpora": corpora, "cases": cases}
    with open(os.path.join(HERE, "golden", "reference_query_generator.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
