"""The request parser against golden fixtures derived WITHOUT it: tests/golden/request_parse.json holds request texts and what
serde's derive(Deserialize) makes of them, restated over Python's json module (tests/reqparse.py).  The product's parser — which the oracle shares —
must understand every text the same way (vq_request_to_json) and reject what serde_json rejects.  Also: the Unicode lowercasing both sides share
(veloci_amd/csrc/text.hpp) against Python's str.lower over every code point."""
import ctypes as C
import json
import os
import unicodedata

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, "golden", "request_parse.json"), encoding="utf-8") as f:
        return json.load(f)["cases"]


def _parse(L, text):
    h = C.c_void_p()
    raw = text.encode("utf-8")
    rc = L.vq_request_parse(raw, len(raw), C.byref(h))
    if rc != 0:
        return rc, L.vq_last_error().decode("utf-8", "replace")
    try:
        return 0, L.vq_request_to_json(h).decode("utf-8")
    finally:
        L.vq_request_free(h)


def test_fixture_is_reproducible_from_the_restatement():
    import reqparse
    for c in _cases():
        if "error" in c:
            with pytest.raises(reqparse.ParseError):
                reqparse.canonical(c["text"])
        else:
            assert reqparse.canonical(c["text"]) == c["parsed"]


def test_parser_understands_every_fixture_as_serde_does():
    from veloci_amd import _lib
    L = _lib.lib()
    cases = _cases()
    assert len(cases) > 120 and sum(1 for c in cases if "error" in c) >= 25
    for c in cases:
        rc, out = _parse(L, c["text"])
        if "error" in c:
            assert rc == 7, (c["text"], rc, out)  # VQ_ERR_JSON
        else:
            assert rc == 0, (c["text"], out)
            assert out == c["parsed"], c["text"]


def test_lowercasing_agrees_with_an_independent_implementation_on_every_code_point():
    """str::to_lowercase (search_field.rs:284,312) as text.hpp implements it vs Python's str.lower — both follow UnicodeData's simple mappings plus
    SpecialCasing's unconditional U+0130.  Differences are allowed only where the two Unicode versions differ (listed, not silently skipped) and for
    the context-sensitive final sigma, which text.hpp documents it does not reproduce."""
    from veloci_amd import _lib
    L = _lib.lib()
    buf = C.create_string_buffer(64)

    def low(s):
        raw = s.encode("utf-8")
        n = L.vq_debug_to_lowercase(raw, len(raw), buf, 64)
        assert n != C.c_size_t(-1).value
        return buf.raw[:n].decode("utf-8")

    differ = []
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        ch = chr(cp)
        if low(ch) != ch.lower():
            differ.append(cp)
    # code points whose lowercase mapping this interpreter's Unicode tables (unicodedata.unidata_version) do not have yet, or have since gained
    assert all(unicodedata.category(chr(cp)) in ("Cn", "Lu", "Lt", "Ll", "Lo") for cp in differ) and len(differ) <= 64, [hex(c) for c in differ[:80]]
    assert low("AbÇ ΔΣ İ") in ("abç δσ i̇", "abç δς i̇")  # final sigma: Rust lowercases a word-final Σ to ς; text.hpp always gives σ
    assert low("STRASSE ẞ Ǆ") == "strasse ß ǆ"


def test_normalize_text_agrees_with_the_reference_regexes_run_by_an_independent_engine():
    """util::normalize_text (src/util.rs:11-29) as text.hpp restates it (passes over code points) vs the reference's five regular expressions
    run by Python's `re` in the same order.  Rust's `\\s` and str::trim use White_Space; Python's `\\s` adds U+001C-001F, so the white-space class
    is spelled out here."""
    import random
    import re
    from veloci_amd import _lib
    L = _lib.lib()
    buf = C.create_string_buffer(4096)
    ws = "\\t\\n\\x0b\\x0c\\r \\x85\\xa0\\u1680\\u2000-\\u200a\\u2028\\u2029\\u202f\\u205f\\u3000"
    steps = [(re.compile(r"\([fmn\d]\)"), " "), (re.compile(r"[\(\)]"), " "), (re.compile("[{}'\"“]"), ""), (re.compile("[%s][%s]+" % (ws, ws)), " "),
             (re.compile("[,.…;・’-]"), "")]
    strip = re.compile("^[%s]+|[%s]+$" % (ws, ws))

    def want(s):
        for rx, rep in steps:
            s = rx.sub(rep, s)
        return strip.sub("", s.lower())

    def got(s):
        raw = s.encode("utf-8")
        n = L.vq_debug_normalize_text(raw, len(raw), buf, 4096)
        assert n != C.c_size_t(-1).value
        return buf.raw[:n].decode("utf-8")

    fixed = ["majestätischer Anblick (m)", "Majestät (f)", "(1) eins", "((f))", "(f", "a  b\t\tc \n d", "it's {so} \"quoted\" “x", "e.g., this; that…・’-", " padded ", "",
             "(٣)", "(５)", "a　　b", " x ", "(f)(m)(n)(9)", "A-B", "Straße (n) ", "(x)", "( f )"]
    for s in fixed:
        assert got(s) == want(s), repr(s)
    rng = random.Random(5)
    alphabet = list("()fmn019 {}'\"“\t\n ,.…;・’-abcXYZäÖ 　٣")
    for _ in range(3000):
        s = "".join(rng.choice(alphabet) for _ in range(rng.randrange(0, 14)))
        assert got(s) == want(s), repr(s)


def test_id_list_sort_agrees_with_numpy_in_all_three_regimes():
    """sort_unique_u32 (compile.cpp): short lists (std::sort), dense spans (bitmap), sparse spans (LSD radix passes) — against np.unique."""
    import numpy as np
    from veloci_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    regimes = set()
    for it in range(120):
        n = int(rng.integers(0, 3000)) if it < 30 else int(rng.integers(2048, 300_000))
        span = [1_000_000, 4_000_000_000, 3 * n + 10, 40 * n + 10][it % 4]
        v = (rng.integers(0, span, n, dtype=np.uint64) + int(rng.integers(0, 1000))).astype(np.uint32)
        want = np.unique(v)
        buf = np.ascontiguousarray(v)
        m = L.vq_debug_sort_unique_u32(buf.ctypes.data_as(C.c_void_p), n)
        assert m == len(want) and np.array_equal(buf[:m], want), (it, n, span)
        if n >= 2048:
            regimes.add("bitmap" if (int(v.max()) - int(v.min()) + 64) // 64 <= n else "radix")
    assert regimes == {"bitmap", "radix"}
