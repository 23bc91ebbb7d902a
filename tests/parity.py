"""Shared helpers of the parity tests: compare a veloci_amd SearchResult with the CPU oracle's."""
import json

import numpy as np

REL_TOL = 1e-5  # BASELINE.json north_star: f32 scores within 1e-5 relative; doc-id sets bit-exact


def assert_same(req, got, want, exact_scores=True):
    ctx = json.dumps(req)[:300]
    assert got.num_hits == want.num_hits, f"num_hits {got.num_hits} != {want.num_hits} for {ctx}"
    assert got.ids.tolist() == want.ids.tolist(), f"ids differ for {ctx}\n got  {got.ids.tolist()}\n want {want.ids.tolist()}\n got  {got.scores.tolist()}\n want {want.scores.tolist()}"
    if exact_scores:
        assert np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)), f"scores not bit-exact for {ctx}\n got  {got.scores.tolist()}\n want {want.scores.tolist()}"
    else:
        np.testing.assert_allclose(got.scores, want.scores, rtol=REL_TOL, atol=0, err_msg=ctx)
    gf = got.facets or {}
    wf = dict(want.facets or [])
    assert set(gf) == set(wf), f"facet fields differ for {ctx}: {set(gf)} vs {set(wf)}"
    for field in wf:
        # the reference leaves the order of equal counts unspecified (facet.rs:19-23 sort_unstable_by):
        # compare the count sequence, and the value set of every count that is fully inside the window
        g, w = gf[field], wf[field]
        assert [c for _, c in g] == [c for _, c in w], f"facet counts differ for {field} in {ctx}: {g} vs {w}"
        if g:
            last = g[-1][1]
            for cnt in set(c for _, c in g):
                if cnt != last:
                    assert {v for v, c in g if c == cnt} == {v for v, c in w if c == cnt}, f"facet values differ for {field} count {cnt}: {g} vs {w}"
