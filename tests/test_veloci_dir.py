"""veloci_amd/veloci_dir.py — the part of a veloci index directory that can be read without the un-vendored crates: metaData.json
-> which vq_index_add_* takes every index, the byte-packed 1:1 arrays (src/indices/direct/single_array.rs) and the heads of 1:n stores
(src/indices/indirect/indirect.rs).  Pinned by the reference's own unit-test vector for the packed arrays; round trips over the mini-indexer's corpora."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from veloci_amd import veloci_dir as vd  # noqa: E402


def test_packed_single_array_matches_the_reference_unit_test():
    """src/indices/direct/single_array.rs:66-92: [123, 33, 545, 99] written with the width of its maximum decodes to 122, 32, 544, 98 (stored = value + 1),
    keys behind the end have no value; [50001, 33] needs three bytes per key"""
    assert [vd.bytes_required(v) for v in (0, 127, 128, 545, 32767, 32768, 50001, 8388607, 8388608)] == [1, 1, 2, 2, 2, 3, 3, 3, 4]
    raw = np.array([123, 33, 545, 99], "<u2").tobytes()
    present, values = vd.decode_single_array(raw, 545)
    assert present.tolist() == [True] * 4 and values.tolist() == [122, 32, 544, 98]
    raw = b"".join(int(v).to_bytes(3, "little") for v in (50001, 33))
    present, values = vd.decode_single_array(raw, 50001)
    assert values.tolist() == [50000, 32]
    rng = np.random.default_rng(5)
    for maxv in (3, 200, 70000, 2 ** 24 + 5):
        vals = rng.integers(0, maxv + 1, 500).astype(np.uint32)
        pres = rng.random(500) < 0.7
        p2, v2 = vd.decode_single_array(vd.encode_single_array(vals, pres, maxv), maxv)
        assert np.array_equal(p2, pres) and np.array_equal(v2[pres], vals[pres]) and not v2[~pres].any()


def test_indirect_heads():
    """indirect/mod.rs:12-20 + create_indirect.rs:60-72: 0 = no values, high bit = the one value inlined, otherwise an offset into .data"""
    kind, val = vd.decode_indirect_heads(np.array([0, (1 << 31) | 7, 12, (1 << 31)], "<u4").tobytes())
    assert kind.tolist() == [0, 1, 2, 1] and val.tolist() == [0, 7, 12, 0]
    raw = vd.encode_indirect_inline([[5], [], [0], [99]])
    kind, val = vd.decode_indirect_heads(raw)
    assert kind.tolist() == [1, 0, 1, 1] and val[kind == 1].tolist() == [5, 0, 99]


def test_directory_round_trip_over_the_reference_corpora(tmp_path):
    """Every index of the mini-indexer's corpora is listed in metaData.json under the category / cardinality persistence.rs:206-291 dispatches on;
    what the module says it can read comes back equal to the arrays that were written, the rest is reported with the crate that owns its format."""
    import refcases
    fx = refcases.load()
    seen_loaded = seen_blocked = 0
    for name in sorted(fx["corpora"])[:6]:
        data, docs, info = refcases.build(name)
        d = str(tmp_path / name)
        written = vd.write_fixture_directory(d, data, data.num_anchors)
        meta, entries = vd.plan(d)
        assert meta["num_docs"] == data.num_anchors
        by_path = {e["path"]: e for e in entries}
        for path in data.token_to_anchor_score:
            assert by_path[path]["adder"] == "vq_index_add_token_to_anchor_score" and by_path[path]["readable"].startswith("no: ")
        for path in data.phrase_pair_to_anchor:
            assert by_path[path]["adder"] == "vq_index_add_phrase_pair_to_anchor"
        for path in data.fst:
            assert by_path[path]["adder"] == "vq_index_add_fst" and "fst" in by_path[path]["readable"]
        got, report = vd.load(d)
        assert all(got.columns[c] == v for c, v in data.columns.items())  # (metaData.json also lists the id-relation paths as columns)
        for e in report:
            if e["loaded"]:
                seen_loaded += 1
                if e["category"] == "Boost":
                    kb, pres, bits = data.boost[e["path"]]
                    gkb, gpres, gbits = got.boost[e["path"]]
                    pres = np.ones(len(bits), np.uint8) if pres is None else pres
                    n = len(bits)
                    assert np.array_equal(gpres[:n], pres) and np.array_equal(gbits[:n][pres.astype(bool)], bits[pres.astype(bool)])
                else:
                    kb, off, vals = data.key_value_stores[e["path"]]
                    gkb, goff, gvals = got.key_value_stores[e["path"]]
                    # (trailing keys without values are not stored by either format)
                    n = min(len(off), len(goff))
                    assert np.array_equal(goff[:n], off[:n]) and np.array_equal(gvals, vals) and (np.diff(off[n - 1:].astype(np.int64)) == 0).all()
            else:
                seen_blocked += 1
                assert e.get("why"), e
        assert set(written) <= {e["path"] for e in report if e["loaded"]}
    assert seen_loaded >= 10 and seen_blocked >= 10
