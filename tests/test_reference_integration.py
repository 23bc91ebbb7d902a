"""The reference's integration tests (tests/all/*.rs), replayed: the mini-indexer builds each test's index from the
test's own documents, then every request must produce what the reference test asserts.  CPU half: through the oracle
(pins the oracle AND the mini-indexer to the reference's expectations).  GPU half: test_gpu_parity.py."""
import json

import numpy as np
import pytest

import refcases

CASES = refcases.load()["cases"]


def _oracle_runner(name):
    from oracle import binding as O
    data, docs, info = refcases.build(name)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    return ora, docs, info


_ORACLES = {}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_reference_assertions(case):
    if case["corpus"] not in _ORACLES:
        _ORACLES[case["corpus"]] = _oracle_runner(case["corpus"])
    ora, docs, info = _ORACLES[case["corpus"]]
    refcases.check_expectations(case, docs, info, lambda req: ora.search_json(json.dumps(req)))


def test_mini_indexer_token_score_matches_oracle():
    from oracle import binding as O
    from veloci_amd import mini_indexer
    for pos in list(range(0, 40)) + [100, 1000]:
        for occ in (1, 2, 3, 10, 100, 5000, 10**6):
            for ntok in (1, 2, 3, 7, 16, 100, 1000):
                for exact in (False, True):
                    assert mini_indexer.token_score(pos, occ, ntok, exact) == O.calculate_token_score(pos, occ, ntok, exact)


def test_tokenizer_vectors():
    # src/tokenizer/mod.rs:41-76 (the reference's tokenizer unit tests)
    from veloci_amd.mini_indexer import tokenize, DEFAULT_SEPARATORS as D
    assert [t for t, _ in tokenize("das \n ist ein txt, test", D)] == ["das", " \n ", "ist", " ", "ein", " ", "txt", ", ", "test"]
    assert [t for t, _ in tokenize(" Taschenbuch (kartoniert)", D)] == [" ", "Taschenbuch", " (", "kartoniert", ")"]
    assert [t for t, _ in tokenize("T oll", D)] == ["T", " ", "oll"]


# ---------------------------------------------------------------- suggest / regex leaves
def _load_suggest_regex():
    import os
    with open(os.path.join(refcases.HERE, "golden", "reference_suggest_regex.json"), encoding="utf-8") as f:
        return json.load(f)


def build_fixture_corpus(fx, name, token_values=None):
    """IndexData + docs of a corpus named by a fixture (its own corpora first, then reference_integration.json's)."""
    from veloci_amd import mini_indexer
    c = fx["corpora"].get(name) or refcases.load()["corpora"][name]
    docs = refcases.corpus_docs(c)
    data, info = mini_indexer.build_index(docs, c["indices"], token_values=tuple(token_values) if token_values else None)
    return data, docs, info


def texts_in_score_groups(entries):
    """[(text, score, id)] -> [sorted texts of every run of equal scores]: the reference's final sort is unstable (search_field.rs:189)"""
    groups = []
    for text, score, _ in entries:
        if groups and groups[-1][0] == score:
            groups[-1][1].append(text)
        else:
            groups.append((score, [text]))
    return [sorted(g[1]) for g in groups]


def check_suggest_case(case, entries):
    got = texts_in_score_groups(entries)
    want, i = [], 0
    for g in got:  # cut the expected sequence into the same group sizes
        want.append(sorted(case["expect_texts"][i:i + len(g)]))
        i += len(g)
    assert got == want and i == len(case["expect_texts"]), f"{case['name']}: {[e[0] for e in entries]} != {case['expect_texts']}"


def test_oracle_reproduces_reference_suggest_and_regex_assertions():
    from oracle import binding as O
    fx = _load_suggest_regex()
    for case in fx["suggest"]:
        data, docs, info = build_fixture_corpus(fx, case["corpus"], case.get("token_values"))
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        check_suggest_case(case, ora.suggest_json(json.dumps(case["request"])))
    for case in fx["highlight"]:  # search_field::highlight, tests/all/tests.rs:1009-1085
        data, docs, info = build_fixture_corpus(fx, case["corpus"])
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        assert [t for t, _, _ in ora.highlight_json(json.dumps(case["request"]))] == case["expect_texts"], case["name"]
    for case in fx["term_lookup"]:
        data, docs, info = build_fixture_corpus(fx, case["corpus"])
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        got = sorted(t for t, _, _ in ora.suggest_json(json.dumps(case["request"])))
        assert got == case["expect_terms_sorted_lowercase"], (case["name"], got)
    for case in fx["regex"]:
        data, docs, info = build_fixture_corpus(fx, case["corpus"])
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        res = ora.search_json(json.dumps(case["request"]))
        assert len(res.ids) == case["expect_len"], (case["name"], res.ids)
        if "expect_doc0" in case:
            assert docs[int(res.ids[0])][case["expect_doc0"][0]] == case["expect_doc0"][1], case["name"]


def test_oracle_regex_match_sets_agree_with_python_re():
    """The oracle's regex leaves (std::regex) against an independent engine: Python's `re` with the reference's semantics — unanchored at the
    start, the match must reach the end of the term (search_field.rs:72-83); starts_with: any prefix may match."""
    import re
    from oracle import binding as O
    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    for path in ("meanings.ger[]", "meanings.eng[]"):
        terms = info[path + ".textindex"]["terms"] if path + ".textindex" in info else info[path]["terms"]
        for pat in (".*wil.*", "w.l+e?", "[a-m]+", ".*(ung|ing)", "maje.*", ".*\\(f\\)", "W.*", ".* .*"):
            for ci in (True, False):
                for sw in (False, True):
                    rx = re.compile(pat, re.IGNORECASE if ci else 0)
                    want = sorted(t for t in terms if (rx.search(t) if sw else re.fullmatch("[\\s\\S]*?(?:" + pat + ")", t, re.IGNORECASE if ci else 0)))
                    part = {"terms": [pat], "path": path, "is_regex": True, "ignore_case": ci, "starts_with": sw}
                    got_ids = [i for _, _, i in ora.suggest_json(json.dumps(part))]
                    got = sorted(terms[i] for i in got_ids)
                    assert got == want, (path, pat, ci, sw, got, want)


# ---------------------------------------------------------------- query generator replay
def _load_query_generator():
    import os
    with open(os.path.join(refcases.HERE, "golden", "reference_query_generator.json"), encoding="utf-8") as f:
        return json.load(f)


def test_query_generator_restatement_regenerates_the_committed_requests():
    """tests/qgen.py (query_generator.rs:175-246) against the committed fixture: the requests are reproducible from each test's parameters, and the
    generator's own errors carry the text the reference tests look for (test_query_generator.rs:362-381)."""
    import qgen
    fx = _load_query_generator()
    for case in fx["cases"]:
        c = fx["corpora"][case["corpus"]]
        if "generator_error_contains" in case["expect"]:
            with pytest.raises(qgen.GeneratorError) as e:
                qgen.search_query(c["all_fields"], c["search_fields"], case["params"])
            assert case["expect"]["generator_error_contains"] in str(e.value), case["name"]
            assert case["generator_error"] == str(e.value)
        else:
            assert qgen.search_query(c["all_fields"], c["search_fields"], case["params"]) == case["request"], case["name"]
    # the pieces of the generator the reference unit-tests itself
    assert qgen.get_default_levenshtein("a" * 2, 1, False) == 0 and qgen.get_default_levenshtein("a" * 3, 1, False) == 1  # query_generator.rs:84-99
    assert qgen.get_default_levenshtein("a" * 3, 1, True) == 0 and qgen.get_default_levenshtein("a" * 6, 2, False) == 2
    assert qgen.regex_escape("a.b*c") == "a\\.b\\*c"
    # operators only between operands (query_parser/src/lexer.rs:265-275)
    assert qgen.parse("OR OR", True) == ("bin", ("leaf", "OR"), "or", ("leaf", "OR"))
    assert qgen.parse("OR OR OR", True) == ("bin", ("leaf", "OR"), "or", ("leaf", "OR"))
    assert qgen.parse("AND AND AND", True) == ("bin", ("leaf", "AND"), "and", ("leaf", "AND"))
    assert qgen.parse("ANDand AND", True) == ("bin", ("leaf", "ANDand"), "or", ("leaf", "AND"))
    flat = qgen.simplify({"or": {"queries": [{"or": {"queries": [{"search": 1}, {"search": 2}]}}, {"search": 3}]}})
    assert flat == {"or": {"queries": [{"search": 3}, {"search": 1}, {"search": 2}]}}


def test_oracle_reproduces_reference_query_generator_assertions():
    """Every search-producing case of tests/all/test_query_generator.rs and the query-generator half of test_code_search.rs, through the oracle."""
    from oracle import binding as O
    fx = _load_query_generator()
    oracles = {}
    ran = 0
    for case in fx["cases"]:
        if "request" not in case:
            continue
        if case["corpus"] not in oracles:
            c = fx["corpora"][case["corpus"]]
            data, docs, info = build_fixture_corpus(fx, case["corpus"], c.get("token_values"))
            ora = O.OracleIndex(data.num_anchors)
            data.load_into(ora)
            oracles[case["corpus"]] = (ora, docs, info)
        ora, docs, info = oracles[case["corpus"]]
        res = refcases.check_expectations(case, docs, info, lambda req: ora.search_json(json.dumps(req)))
        if case["request"].get("explain"):  # hits and scores do not depend on the flag
            plain = ora.search_json(json.dumps({k: v for k, v in case["request"].items() if k != "explain"}))
            assert plain.ids.tolist() == res.ids.tolist() and plain.scores.tolist() == res.scores.tolist() and plain.explain == [None] * len(plain.ids)
        ran += 1
    assert ran == 29


# ---------------------------------------------------------------- explain
def _load_explain():
    import os
    with open(os.path.join(refcases.HERE, "golden", "reference_explain.json"), encoding="utf-8") as f:
        return json.load(f)


def explain_requests():
    """Requests that walk every record-producing path of the reference (search_field.rs:334-343, 429-441; set_op.rs:132-137, 187-208, 421-433;
    boost.rs:297-300, 371-374) on the `test_all` corpus — shared by the oracle's own checks and the GPU parity test."""
    leaf = lambda **kw: {"search": kw}
    ger, eng = "meanings.ger[]", "meanings.eng[]"
    reqs = []
    for top in ({}, {"top": 3, "skip": 1}):
        reqs += [
            dict({"search_req": leaf(terms=["majestät"], path=ger, levenshtein_distance=2), "explain": True}, **top),
            dict({"search_req": leaf(terms=["will"], path=eng, starts_with=True, boost=2.5), "explain": True}, **top),
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=ger, levenshtein_distance=1), leaf(terms=["will"], path=eng), leaf(terms=["urge"], path=eng)]}}, "explain": True}, **top),
            dict({"search_req": {"and": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["urge"], path=eng)]}}, "explain": True}, **top),
            dict({"search_req": {"and": {"queries": [leaf(terms=["urge"], path=eng), leaf(terms=["will"], path=eng, levenshtein_distance=1), leaf(terms=["begeisterung"], path=ger, levenshtein_distance=1)]}},
                  "explain": True}, **top),
            dict({"search_req": {"or": {"queries": [{"and": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["urge"], path=eng)]}}, leaf(terms=["majestät"], path=ger),
                                                    {"or": {"queries": [leaf(terms=["anblick"], path=ger, levenshtein_distance=1), leaf(terms=["will"], path=eng)]}}]}}, "explain": True}, **top),
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["majestät"], path=ger)]}}, "explain": True,
                  "boost": [{"path": "commonness", "boost_fun": "Log10", "param": 1}, {"path": "commonness", "boost_fun": "Multiply", "skip_when_score": [3.0]}]}, **top),
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["majestät"], path=ger)]}}, "explain": True,
                  "filter": {"search": {"terms": ["nice"], "path": "tags[]"}}, "boost_term": [{"terms": ["will"], "path": eng, "boost": 3.0}], "text_locality": True}, **top),
            dict({"search_req": leaf(terms=[".*e.*"], path=ger, is_regex=True), "explain": True, "why_found": True}, **top),
        ]
    return reqs


def highlight_parts():
    """RequestSearchParts for search_field::highlight on the `test_all` corpus: every branch of resolve_token_hits_to_text_id (:550-639) and
    highlight_document (highlight_field.rs:187-272) — several windows, windows that touch, ellipses on either side, hits at the text's edges,
    snippet options, terms that util::normalize_text rewrites, and the requests on which the reference panics.  Shared by the oracle's own
    checks, the sanitizer driver (the exact-term ones) and the GPU parity test."""
    long_paths = ("mylongtext", "tags[]", "sub_level[].text")
    parts = []
    for path in long_paths:
        for term in ("story", "Story,", "(the)", "a", "the", "prolog", "end", "world", "guy", "rule", "nope"):
            for kw in ({}, {"starts_with": True}, {"levenshtein_distance": 1}):
                parts.append(dict({"terms": [term], "path": path, "snippet": True}, **kw))
        for si in ({"num_words_around_snippet": 0}, {"num_words_around_snippet": 1}, {"num_words_around_snippet": 2, "snippet_connector": " [..] "},
                   {"num_words_around_snippet": 3, "max_snippets": 1}, {"max_snippets": 0}, {"snippet_start_tag": "<em>", "snippet_end_tag": "</em>"},
                   {"num_words_around_snippet": 40}, {"num_words_around_snippet": -1}):
            for term, kw in (("the", {}), ("a", {}), ("t", {"starts_with": True}), ("wen", {"levenshtein_distance": 1})):
                parts.append(dict({"terms": [term], "path": path, "snippet": True, "snippet_info": si}, **kw))
    for path in ("meanings.ger[]", "meanings.eng[]"):
        for term, kw in (("majestät", {}), ("Majestät (f)", {}), ("anblick", {"levenshtein_distance": 1}), ("will", {"starts_with": True}), ("will", {"starts_with": True, "top": 2, "skip": 1}),
                         ("test", {"boost": -2.0}), ("der", {"top": 1}), ("ist", {"ignore_case": False}), ("treffer", {"levenshtein_distance": 2, "top": 3})):
            parts.append(dict({"terms": [term], "path": path, "snippet": True}, **kw))
    parts += [
        {"terms": ["story"], "path": "mylongtext"},                                  # no "snippet": the token hits keep no snippet -> the reference panics
        {"terms": ["story"], "path": "mylongtext", "snippet": False},
        {"terms": ["nope"], "path": "mylongtext"},                                   # nothing matched: an empty result
        {"terms": ["Prolog:\nthis is a story of a guy who went out to rule the world, but then died. the end"], "path": "mylongtext", "snippet": True},
        {"terms": ["nice"], "path": "tags[]", "snippet": True},                      # an untokenised-looking text: a text that is its own only token has no token rows
        {"terms": ["1587690"], "path": "ent_seq", "snippet": True},
        {"terms": ["story"], "path": "nosuchfield", "snippet": True},
        {"terms": [], "path": "mylongtext", "snippet": True},
        {"terms": ["(f)"], "path": "meanings.ger[]", "snippet": True},               # normalises to the empty term
    ]
    return parts


def test_oracle_highlight_invariants():
    """The oracle's highlight on the wider request set: every returned snippet, with its tags and connectors removed, is made of pieces of the
    stored text; every tagged token is one of the matched terms; scores come out ranked."""
    from oracle import binding as O
    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    answered = errors = 0
    for part in highlight_parts():
        try:
            got = ora.highlight_json(json.dumps(part))
        except O.OracleError:
            errors += 1
            continue
        answered += 1
        scores = [s for _, s, _ in got]
        assert scores == sorted(scores, reverse=True), part
        si = part.get("snippet_info", {})
        st, en, conn = si.get("snippet_start_tag", "<b>"), si.get("snippet_end_tag", "</b>"), si.get("snippet_connector", " ... ")
        key = part["path"] + ".textindex"
        kb, off, vals = data.key_value_stores[key + ".text_id_to_token_ids"]
        terms = info[part["path"]]["terms"]
        for snippet, _, text_id in got:
            row = vals[int(off[text_id - kb]):int(off[text_id - kb + 1])]
            full = "".join(terms[t] for t in row)
            for piece in snippet.split(conn):  # (windows of hits 10-20 tokens apart overlap in the reference: a piece may start before the last one ended)
                assert piece.replace(st, "").replace(en, "") in full, (part, snippet)
            if si.get("max_snippets", 1) != 0:
                assert st in snippet and en in snippet, (part, snippet)
    assert answered > 150 and errors > 10, (answered, errors)


def test_oracle_reproduces_reference_explain_assertions_and_its_own_invariants():
    from oracle import binding as O
    fx = _load_explain()
    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    for case in fx["cases"]:
        refcases.check_expectations(case, docs, info, lambda req: ora.search_json(json.dumps(req)))
    saw = set()
    for req in explain_requests():
        res = ora.search_json(json.dumps(req))
        plain = ora.search_json(json.dumps({k: v for k, v in req.items() if k != "explain"}))
        assert plain.ids.tolist() == res.ids.tolist() and plain.scores.tolist() == res.scores.tolist(), req
        assert len(res.explain) == len(res.ids)
        for recs in res.explain:
            for r in recs or []:
                kind = next(iter(r))
                saw.add(kind)
                if kind == "TermToAnchor":  # search_field.rs:426
                    t = r[kind]
                    assert np.float32(t["term_score"]) * np.float32(t["anchor_score"]) == np.float32(t["final_score"])
    assert saw == {"TermToAnchor", "LevenshteinScore", "OrSumOverDistinctTerms", "Boost"}
    # a plain single leaf: the hit's score IS the largest final_score among its records
    res = ora.search_json(json.dumps({"search_req": {"search": {"terms": ["will"], "path": "meanings.eng[]", "levenshtein_distance": 1}}, "explain": True}))
    for score, recs in zip(res.scores, res.explain):
        assert np.float32(max(r["TermToAnchor"]["final_score"] for r in recs if "TermToAnchor" in r)) == score


# ---------------------------------------------------------------- highlight_text (why_found highlighting of a document's texts)
def test_highlight_text_matches_the_reference_vectors_the_oracle_and_a_python_restatement():
    """highlight_field::highlight_text (highlight_field.rs:92-146), host work with no index behind it: the reference's four assertions (:279-320)
    on the oracle and on the product (`vq_highlight_text`), then product == oracle == a restatement over mini_indexer's tokenizer (itself pinned by the
    reference's tokenizer vectors) on random texts, term sets and snippet options."""
    import random
    import veloci_amd
    from oracle import binding as O
    from veloci_amd.mini_indexer import tokenize, DEFAULT_SEPARATORS as D
    fx = _load_suggest_regex()
    for case in fx["highlight_text"]:
        assert O.highlight_text(case["text"], case["terms"]) == case["expect"], case
        assert veloci_amd.highlight_text(case["text"], case["terms"]) == case["expect"], case

    def restated(text, terms, si, tokenized):
        st, en, conn = si.get("snippet_start_tag", "<b>"), si.get("snippet_end_tag", "</b>"), si.get("snippet_connector", " ... ")
        around, max_snippets = 2 * si.get("num_words_around_snippet", 5), si.get("max_snippets", 2**32 - 1)
        terms = set(terms)
        if len(terms) == 1 and text in terms:
            return st + text + en
        if not tokenized:
            return None
        toks = [t for t, _ in tokenize(text, D)]
        hits = [i for i, t in enumerate(toks) if t in terms]
        groups, prev = [], -around
        for h in hits:
            if h - prev >= around:
                groups.append([])
            prev = h
            groups[-1].append(h)
        parts = ["".join(st + toks[i] + en if toks[i] in terms else toks[i] for i in range(max(g[0] - around, 0), min(g[-1] + around + 1, len(toks)))) for g in groups[:max_snippets]]
        out = conn.join(parts)
        if hits and hits[0] > around:
            out = conn + out
        if hits and hits[-1] < len(toks) - around:
            out += conn
        return out if parts else None

    rng = random.Random(11)
    words = ["story", "guy", "the", "a", "Schön", "Hans", "treffer", "went", "rule", "world", "end", "died", "Prolog", "意欲", "ok"]
    seps = [" ", ", ", ".", " - ", "\n", ": ", "(", ") ", "…", "—", "\t", "'", "™ "]
    for _ in range(1500):
        n = rng.randrange(0, 40)
        text = "".join(rng.choice(words) + rng.choice(seps) for _ in range(n))
        if rng.random() < 0.3:
            text = rng.choice(seps) + text
        if rng.random() < 0.3 and text:
            text = text.rstrip(" ,.-\n:()…—\t'™")
        terms = rng.sample(words, rng.randrange(0, 4)) + ([text] if rng.random() < 0.05 else []) + ([", "] if rng.random() < 0.05 else [])
        si = {}
        if rng.random() < 0.5:
            si = {"num_words_around_snippet": rng.randrange(0, 5)}
            if rng.random() < 0.5:
                si["max_snippets"] = rng.randrange(0, 3)
            if rng.random() < 0.3:
                si.update(snippet_start_tag="[", snippet_end_tag="]", snippet_connector=" ~ ")
        tokenized = rng.random() < 0.9
        want = restated(text, terms, si, tokenized)
        assert O.highlight_text(text, terms, si or None, tokenized) == want, (text, terms, si, tokenized)
        assert veloci_amd.highlight_text(text, terms, si or None, tokenized) == want, (text, terms, si, tokenized)
    with pytest.raises(veloci_amd.VelociError):
        veloci_amd.highlight_text("a b", ["a"], {"num_words_around_snippet": -1})
    with pytest.raises(veloci_amd.VelociError):
        veloci_amd.highlight_text("a b", ["a"], {"max_snippets": "x"})


# ---------------------------------------------------------------- why_found with select (search.rs:220-224, search/why_found.rs)
def _load_why_found():
    import os
    with open(os.path.join(refcases.HERE, "golden", "reference_why_found.json"), encoding="utf-8") as f:
        return json.load(f)


def why_found_select_requests():
    """Requests with `why_found` and `select` on the reference's `test_all` corpus: one and several fields, 1:n and nested paths, fuzzy / prefix terms
    (several matched tokens per text), trees, top / skip windows, a field searched twice.  Shared by the oracle's own checks and the GPU parity test."""
    def leaf(**kw):
        return {"search": kw}
    reqs = []
    for path in ("mylongtext", "tags[]", "sub_level[].text", "meanings.ger[]", "meanings.eng[]", "ent_seq", "field1[].text"):
        for term, kw in (("story", {}), ("the", {}), ("a", {}), ("nice", {}), ("majestät", {}), ("will", {"starts_with": True}), ("test", {"levenshtein_distance": 1}),
                         ("der", {}), ("1587690", {}), ("awesome", {})):
            reqs.append({"search_req": leaf(terms=[term], path=path, **kw), "why_found": True, "select": ["ent_seq"]})
    reqs += [
        {"search_req": {"or": {"queries": [leaf(terms=["story"], path="mylongtext"), leaf(terms=["nice"], path="tags[]"), leaf(terms=["will"], path="meanings.eng[]")]}},
         "why_found": True, "select": ["tags[]"], "top": 20},
        {"search_req": {"and": {"queries": [leaf(terms=["the"], path="mylongtext"), leaf(terms=["a"], path="mylongtext")]}}, "why_found": True, "select": []},
        {"search_req": {"or": {"queries": [leaf(terms=["the"], path="mylongtext"), leaf(terms=["the"], path="mylongtext", levenshtein_distance=1)]}},
         "why_found": True, "select": ["x"]},
        {"search_req": {"or": {"queries": [leaf(terms=["majestät"], path="meanings.ger[]"), leaf(terms=["majestätischer"], path="meanings.ger[]"),
                                           leaf(terms=["anblick"], path="meanings.ger[]", levenshtein_distance=1)]}}, "why_found": True, "select": ["meanings"], "top": 3, "skip": 1},
        {"search_req": leaf(terms=["will"], path="meanings.eng[]", starts_with=True), "why_found": True, "select": ["a"], "top": 2, "skip": 2},
        {"search_req": leaf(terms=["nope"], path="mylongtext"), "why_found": True, "select": ["a"]},
        {"search_req": leaf(terms=["story"], path="mylongtext"), "why_found": True, "select": ["a"], "top": 0},
        {"search_req": leaf(terms=["story"], path="mylongtext"), "why_found": True, "select": ["a"], "text_locality": True},
        {"search_req": leaf(terms=["nice"], path="tags[]"), "why_found": True, "select": ["a"], "filter": leaf(terms=["cool"], path="tags[]"),
         "facets": [{"field": "tags[]"}]},
        # what search::search does not look at changes nothing: snippet options on a leaf (plan_steps.rs:184 resolves ids only), a suggest beside the search
        {"search_req": leaf(terms=["story"], path="mylongtext", snippet=True, snippet_info={"max_snippets": 1}), "why_found": True, "select": ["a"]},
        {"search_req": {"or": {"queries": [leaf(terms=["story"], path="mylongtext", snippet=True), leaf(terms=["story"], path="mylongtext")]}}},
        {"search_req": leaf(terms=["will"], path="meanings.eng[]", starts_with=True, snippet_info={"num_words_around_snippet": 1}), "top": 3},
        {"search_req": leaf(terms=["nice"], path="tags[]"), "suggest": [{"terms": ["ni"], "path": "tags[]", "starts_with": True}]},
        {"search_req": leaf(terms=["story"], path="mylongtext"), "select": ["mylongtext"]},   # select without why_found: nothing more than the search
        {"search_req": leaf(terms=["story"], path="mylongtext"), "why_found": True},           # why_found without select: the terms only
    ]
    return reqs


def test_oracle_reproduces_reference_why_found_with_select_assertions():
    from oracle import binding as O
    from veloci_amd import mini_indexer
    fx = _load_why_found()
    oracles = {}
    for case in fx["cases"]:
        if case["corpus"] not in oracles:
            c = fx["corpora"][case["corpus"]]
            data, info = mini_indexer.build_index(refcases.corpus_docs(c), c["indices"])
            ora = O.OracleIndex(data.num_anchors)
            data.load_into(ora)
            oracles[case["corpus"]] = (ora, refcases.corpus_docs(c))
        ora, docs = oracles[case["corpus"]]
        res = ora.search_json(json.dumps(case["request"]))
        assert len(res.ids) >= 1, case["name"]
        for k, v in case.get("expect_doc", {}).items():
            assert docs[int(res.ids[0])][k] == v, case["name"]
        assert res.why_found_info.get(int(res.ids[0]), {}) == case["expect_why_found"], (case["name"], res.why_found_info)
    # wider, on the reference's main test corpus: only returned anchors carry entries, every entry is one of the anchor's texts of the field with
    # its matched tokens tagged, nothing is produced unless both flags are set
    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    with_entries = 0
    for req in why_found_select_requests():
        try:
            res = ora.search_json(json.dumps(req))
        except O.OracleError:
            continue
        if not (req.get("why_found") and "select" in req):
            assert res.why_found_info == {}, req
            continue
        assert set(res.why_found_info) <= set(int(i) for i in res.ids), req
        for anchor, fields in res.why_found_info.items():
            for field, texts in fields.items():
                assert field + ".textindex" in res.why_found_terms, (req, field)
                terms = set(res.why_found_terms[field + ".textindex"])
                for t in texts:
                    with_entries += 1
                    assert "<b>" in t and "</b>" in t, (req, t)
                    tagged = [piece.split("</b>")[0] for piece in t.split("<b>")[1:]]
                    assert all(x in terms for x in tagged), (req, t, terms)
    assert with_entries > 40, with_entries
