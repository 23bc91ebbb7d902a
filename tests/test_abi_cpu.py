"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header declares,
parses requests like serde does, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "veloci_amd.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vq_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import veloci_amd
    from veloci_amd._lib import SYMBOLS
    L = veloci_amd.lib()
    declared = header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/veloci_amd.h but not exported"
    assert set(declared) == set(SYMBOLS), set(declared) ^ set(SYMBOLS)
    assert b"gfx950" in L.vq_version()


def test_integration_guide_binds_every_declared_symbol():
    """INTEGRATION.md shows the reference-side (Rust) extern block a maintainer would add: it must cover the whole header."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        bound = set(re.findall(r"pub fn (vq_[a-z0-9_]+)", f.read()))
    assert bound == set(header_symbols()), bound ^ set(header_symbols())


def test_request_parse_serde_semantics():
    import veloci_amd
    # unknown keys are ignored (tests/all/tests.rs:528 of the reference passes "firstCharExactMatch")
    veloci_amd.Request({"search_req": {"search": {"path": "a", "terms": ["x"], "firstCharExactMatch": True}}, "whatever": 1})
    veloci_amd.Request({"search_req": {"or": {"queries": [{"search": {"path": "a", "terms": ["x"], "levenshtein_distance": 1, "boost": 2.5}}]}},
                        "boost": [{"path": "p", "boost_fun": "Log10", "param": 1, "skip_when_score": [1.0], "expression": "$SCORE + 2.0"}],
                        "facets": [{"field": "f"}, {"field": "g", "top": 3}], "top": 5, "skip": 2, "text_locality": True,
                        "phrase_boosts": [{"search1": {"path": "a", "terms": ["x"]}, "search2": {"path": "a", "terms": ["y"]}}],
                        "filter": {"and": {"queries": [{"search": {"path": "a", "terms": ["z"]}}]}}})
    for bad, code in (("{", 7), ('{"search_req": {"search": {"terms": ["x"]}}}', 7), ('{"search_req": {"xor": {}}}', 7),
                      ('{"search_req": {"search": {"path": "a", "terms": "x"}}}', 7), ('{"boost": [{"path": "p", "boost_fun": "Log3"}]}', 7)):
        with pytest.raises(veloci_amd.VelociError) as e:
            veloci_amd.Request(bad)
        assert e.value.code == code, bad


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (the oracle is never a fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import veloci_amd
    from veloci_amd import synth
    data, _ = synth.generate(synth.SynthSpec(num_docs=2000, num_terms=100, triples=1, with_facets=False, with_boost=False), device="cpu")
    with pytest.raises(veloci_amd.VelociError) as e:
        veloci_amd.Index(data, device=0)
    assert e.value.code == 5 and "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    """Nothing under veloci_amd/ may import, include or link oracle/ (tier rule 3)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "veloci_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                with open(os.path.join(base, fn), errors="ignore") as f:
                    txt = f.read()
                if re.search(r"(from|import)\s+oracle|oracle/|libveloci_oracle|veloci_oracle\.hpp", txt):
                    bad.append(os.path.join(base, fn))
    assert bad == []
