"""Replay helpers for tests/golden/reference_integration.json (inputs and assertions of the reference's integration tests)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def load():
    with open(os.path.join(HERE, "golden", "reference_integration.json"), encoding="utf-8") as f:
        return json.load(f)


def corpus_docs(c):
    return c["docs"] * int(c.get("repeat", 1)) + c.get("tail", [])


def build(name):
    """(IndexData, docs, info) for a corpus, built once per process by the mini-indexer."""
    if name not in _CACHE:
        from veloci_amd import mini_indexer
        c = load()["corpora"][name]
        docs = corpus_docs(c)
        data, info = mini_indexer.build_index(docs, c["indices"])
        _CACHE[name] = (data, docs, info)
    return _CACHE[name]


def dig(doc, path):
    cur = doc
    for p in path:
        cur = cur[p]
    return cur


def check_expectations(case, docs, info, run):
    """run(request_dict) -> object with num_hits, ids, scores, facets (dict or list of pairs); raises on error."""
    exp = case["expect"]
    name = case["name"]
    if "error" in exp:
        try:
            run(case["request"])
        except Exception as e:  # noqa: BLE001
            assert str(e) == exp["error"], f"{name}: error text {e!s}"
            return None
        raise AssertionError(f"{name}: expected error {exp['error']!r}")
    res = run(case["request"])
    ids = [int(i) for i in res.ids]
    if "len" in exp:
        assert len(ids) == exp["len"], f"{name}: hits.len() {len(ids)} != {exp['len']} ({ids})"
    if "num_hits" in exp:
        assert res.num_hits == exp["num_hits"], f"{name}: num_hits {res.num_hits}"
    for i, path, want in exp.get("doc", []):
        assert i < len(ids), f"{name}: no hit {i} ({ids})"
        got = dig(docs[ids[i]], path)
        assert got == want, f"{name}: hits[{i}] is doc {ids[i]} with {path} = {got!r}, the reference asserts {want!r} (ids {ids}, scores {list(res.scores)})"
    if "explain_len0" in exp:  # hits[0].explain.as_ref().unwrap().len()
        assert res.explain and res.explain[0] is not None and len(res.explain[0]) == exp["explain_len0"], f"{name}: hits[0].explain {res.explain[0] if res.explain else None}"
    if "explain_kinds0" in exp:
        assert [next(iter(r)) for r in res.explain[0]] == exp["explain_kinds0"], f"{name}: {res.explain[0]}"
    facets = res.facets
    if facets is not None and not isinstance(facets, dict):
        facets = dict(facets)
    for field, want in exp.get("facets", {}).items():
        got = [[v, int(c)] for v, c in facets[field]]
        assert got == want, f"{name}: facet {field} {got} != {want}"
    for field, want in exp.get("facets_unordered", {}).items():
        got = sorted([v, int(c)] for v, c in facets[field])
        assert got == sorted(want), f"{name}: facet {field} {got} != {want}"
    if "score0_gt_request" in exp:
        other = run(exp["score0_gt_request"])
        assert res.scores[0] > other.scores[0], f"{name}: {res.scores[0]} !> {other.scores[0]}"
    if "score0_gt" in exp:  # assert_gt!(res[0].hit.score, 40.0)
        assert float(res.scores[0]) > exp["score0_gt"], f"{name}: {res.scores[0]} !> {exp['score0_gt']}"
    if "score0_eq_base" in exp:  # assert_eq!(res_unboosted[0].hit.score + 2.0, res_boosted[0].hit.score): an exact f32 equality
        import numpy as np
        e = exp["score0_eq_base"]
        base = np.float32(run(e["request"]).scores[0])
        want = base + np.float32(e["value"]) if e["op"] == "add" else base * np.float32(e["value"])
        assert np.float32(res.scores[0]) == np.float32(want), f"{name}: {res.scores[0]} != {base} {e['op']} {e['value']}"
    if "identity_column" in exp:
        assert info[exp["identity_column"]]["identity"], f"{name}: {exp['identity_column']} is not an identity column"
    return res
