"""Pins the CPU oracle against the known-answer vectors of the reference's own unit tests
(tests/golden/reference_unit_vectors.json)."""
import json
import os

import numpy as np
import pytest

from oracle import binding as O


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "reference_unit_vectors.json")) as f:
        return json.load(f)


def _pairs(x):
    return [(int(a), float(np.float32(b))) for a, b in x]


def test_intersect_hits_score(vec):
    for case in vec["intersect_hits_score"]:
        got = O.intersect_hits_score([_pairs(l) for l in case["lists"]])
        if "expect" in case:
            assert got == _pairs(case["expect"]), case["source"]
        else:
            assert [g[0] for g in got] == case["expect_ids"], case["source"]


def test_intersect_hits_ids(vec):
    for case in vec["intersect_hits_ids"]:
        assert O.intersect_hits_ids(case["lists"]) == case["expect"], case["source"]


def test_union_hits_ids(vec):
    for case in vec["union_hits_ids"]:
        assert O.union_hits_ids(case["lists"]) == case["expect"], case["source"]


def test_intersect_score_hits_with_ids(vec):
    for case in vec["intersect_score_hits_with_ids"]:
        assert O.intersect_score_hits_with_ids(_pairs(case["hits"]), case["ids"]) == _pairs(case["expect"]), case["source"]


def test_intersect_score_hits_with_empty_ids_keeps_hits():
    # set_op.rs:316 — `if let Some(first)`: an empty id list leaves the scored hits untouched
    hits = [(1, 1.0), (2, 2.0)]
    assert O.intersect_score_hits_with_ids(hits, []) == hits


def test_union_hits_score(vec):
    for case in vec["union_hits_score"]:
        got = O.union_hits_score([_pairs(l) for l in case["lists"]], case["terms"])
        assert got == _pairs(case["expect"]), case["source"]


def test_union_same_term_shares_slot():
    # set_op.rs:143,176: the same term string in two operands shares one slot (max, n = 1)
    got = O.union_hits_score([[(1, 3.0), (2, 1.0)], [(1, 2.0), (3, 4.0)]], ["a", "a"])
    assert got == [(1, 3.0), (2, 1.0), (3, 4.0)]


def test_boost_hits_ids_vec_multi(vec):
    for case in vec["boost_hits_ids_vec_multi"]:
        got = O.boost_hits_ids_vec_multi(_pairs(case["hits"]), case["boost_lists"])
        assert got == _pairs(case["expect"]), case["source"]


def test_apply_boost_values_anchor(vec):
    for case in vec["apply_boost_values_anchor"]:
        got = O.apply_boost_values_anchor(_pairs(case["hits"]), _pairs(case["boosts"]), boost_fun=case["boost_fun"])
        assert got == _pairs(case["expect"]), case["source"]


def test_distance(vec):
    for case in vec["distance"]:
        for a, b, d in case["cases"]:
            assert O.distance(a, b) == d, (case["source"], a, b)
            assert O.levenshtein(a, b) == d


def test_osa_vs_levenshtein():
    assert O.levenshtein("ab", "ba", transposition=False) == 2
    assert O.levenshtein("ab", "ba", transposition=True) == 1
    assert O.levenshtein("Haus", "haus", ci=True) == 0
    assert O.levenshtein("Haus", "haus", ci=False) == 1
    assert O.levenshtein("ÄRGER", "ärger", ci=True) == 0


def test_score_expression(vec):
    for case in vec["score_expression"]:
        for expr, rank, want in case["cases"]:
            assert O.score_expression(expr, rank) == want, (case["source"], expr)


def test_f16_roundtrip(vec):
    for case in vec["f16_roundtrip"]:
        for v in case["values"]:
            assert O.f16_to_f32(O.f32_to_f16(float(v))) == float(v), v
    # RNE beyond 2048: 2049 -> 2048, 2051 -> 2052
    assert O.f16_to_f32(O.f32_to_f16(2049.0)) == 2048.0
    assert O.f16_to_f32(O.f32_to_f16(2051.0)) == 2052.0
    # against numpy's IEEE binary16
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-70000, 70000, 2000), rng.uniform(-1e-4, 1e-4, 2000), [0.0, 65504.0, 65520.0, 1e-8]]).astype(np.float32)
    for x in xs:
        want = np.float16(x)
        got = O.f32_to_f16(float(x))
        assert got == int(want.view(np.uint16)), x
        if np.isfinite(want):
            assert O.f16_to_f32(got) == float(want)


def test_default_score_for_distance():
    # search_field.rs:27-33; values listed in SURVEY.md A1
    assert O.default_score_for_distance(0, False) == 10.0
    assert O.default_score_for_distance(1, False) == pytest.approx(1.6666666, rel=1e-7)
    assert O.default_score_for_distance(2, False) == pytest.approx(0.9090909, rel=1e-7)
    assert O.default_score_for_distance(2, True) == pytest.approx(1.1204717, rel=1e-6)
    assert O.default_score_for_distance(0, True) == 10.0


def test_calculate_token_score_range():
    # create/calculate_score.rs:34-49; SURVEY.md A0: ~140-150 for a token, ~395 for an exact whole-text match
    assert 380 <= O.calculate_token_score(0, 1, 1, True) <= 400
    s = O.calculate_token_score(0, 1, 4, False)
    assert 100 <= s <= 160
    assert O.calculate_token_score(15, 100000, 16, False) < s


def test_top_n_sort_order_and_ties():
    # sort.rs:5-22 + search.rs:122-130: score desc, then id DESC
    hits = [(1, 5.0), (2, 7.0), (3, 5.0), (4, 1.0)]
    assert O.top_n_sort(hits, 10) == [(2, 7.0), (3, 5.0), (1, 5.0), (4, 1.0)]
    # more than top_n + 200 entries: still an exact top-n prefix
    rng = np.random.default_rng(3)
    big = [(i, float(np.float32(rng.integers(0, 50)))) for i in range(2000)]
    got = O.top_n_sort(big, 10)[:10]
    want = sorted(big, key=lambda h: (-h[1], -h[0]))[:10]
    assert got == want


def test_steps(vec):
    for case in vec["steps"]:
        for path, want in case["cases"]:
            assert O.steps_to_anchor(path) == want, (case["source"], path)
