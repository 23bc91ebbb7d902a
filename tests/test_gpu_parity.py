"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded synthetic corpora.
Doc-id lists bit-exact; f32 scores bit-exact wherever only + * / are involved, 1e-5 relative for log boosts."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def corpus():
    import veloci_amd
    from veloci_amd import synth
    from oracle import binding as O
    spec = synth.SynthSpec(num_docs=300_000, num_terms=5000, triples=2, extra_probe_dfs=(1000, 30_000, 300_000), background_terms=40)
    data, meta = synth.generate(spec)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    return data, meta, idx, ora


def check(corpus, req, exact_scores=True):
    import veloci_amd
    from parity import assert_same
    data, meta, idx, ora = corpus
    got = veloci_amd.search(req, idx)
    want = ora.search_json(json.dumps(req))
    assert_same(req, got, want, exact_scores)
    return got


def test_native_library_is_loaded():
    import veloci_amd
    assert veloci_amd.lib_path().endswith("libveloci_amd.so")
    assert b"gfx950" in veloci_amd.lib().vq_version()


def test_single_term(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    for t in meta.extra_probes + list(meta.triples[0]):
        for top in (1, 10, 100):
            r = check(corpus, synth.req_single(t, top=top))
            assert r.num_hits > 0


def test_single_term_skip(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    t = meta.extra_probes[1]
    check(corpus, synth.req_single(t, top=10, skip=5))
    r = check(corpus, synth.req_single(meta.extra_probes[0], top=10, skip=1000))
    check(corpus, synth.req_single(t, top=0))
    assert len(r.ids) == min(10, max(0, r.num_hits - 1000))


def test_unknown_term_and_background(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    r = check(corpus, synth.req_single("zzzzzzzzzzzzzzzzzzzz"))
    assert r.num_hits == 0 and len(r.ids) == 0
    for t in meta.background[:10]:
        check(corpus, synth.req_single(t))


def test_case_insensitive_default(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    t = meta.extra_probes[0]
    check(corpus, synth.req_single(t.upper()))
    check(corpus, {"search_req": {"search": {"path": "body", "terms": [t.upper()], "ignore_case": False}}})


def test_and(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    for tri in meta.triples:
        a, b, c = tri
        for terms in ([a, b], [b, c], [a, b, c], [c, b, a], [b, a, c]):
            r = check(corpus, synth.req_and(terms))
            assert r.num_hits > 0
    check(corpus, synth.req_and([meta.triples[0][0], meta.triples[1][2], meta.extra_probes[2]], top=50))
    check(corpus, synth.req_and([meta.triples[0][0], "zzzzzzzzzzzzzzz"]))


def test_or(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    check(corpus, synth.req_or([a, b]))
    check(corpus, synth.req_or([a, b, c], top=25))
    check(corpus, synth.req_or([c, meta.extra_probes[0], meta.background[3]]))
    check(corpus, synth.req_or([a, a]))  # same term twice: one slot (set_op.rs:143,176)


def test_nested(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    d, e, f = meta.triples[1]
    leaf = lambda t: {"search": {"path": "body", "terms": [t]}}
    check(corpus, {"search_req": {"and": {"queries": [{"or": {"queries": [leaf(a), leaf(d)]}}, {"or": {"queries": [leaf(b), leaf(e)]}}]}}})
    check(corpus, {"search_req": {"or": {"queries": [{"or": {"queries": [leaf(a), leaf(d)]}}, leaf(c)]}}})
    check(corpus, {"search_req": {"and": {"queries": [leaf(a)]}}})
    check(corpus, {"search_req": {"or": {"queries": [{"and": {"queries": [leaf(a), leaf(b), leaf(c)]}}, leaf(f)]}}})


def test_leaf_boost(corpus):
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    check(corpus, {"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": [a], "boost": 2.5}}, {"search": {"path": "body", "terms": [b], "boost": 0.3}}]}}})


def test_phrase_and_locality(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    for tri in meta.triples:
        check(corpus, synth.req_and_phrase_locality(list(tri)))
    a, b, c = meta.triples[0]
    r = synth.req_or([a, b, c])
    r["text_locality"] = True
    check(corpus, r)
    r2 = synth.req_and([a, b])
    r2["phrase_boosts"] = [{"search1": {"path": "body", "terms": [a]}, "search2": {"path": "body", "terms": [b]}}]
    check(corpus, r2)


def test_column_boosts(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    for fun, exact in (("Multiply", True), ("Add", True), ("Replace", True), ("Log10", False), ("Log2", False)):
        r = synth.req_and([a, b])
        r["boost"] = [{"path": "pop", "boost_fun": fun, "param": 1.0}]
        check(corpus, r, exact_scores=exact)
    r = synth.req_and([a, b])
    r["boost"] = [{"path": "pop", "boost_fun": "Add", "expression": "$SCORE * 2.0"}, {"path": "pop", "boost_fun": "Multiply", "skip_when_score": [20.6]}]
    check(corpus, r)


def test_and_of_ors_full(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    d, e, f = meta.triples[1]
    check(corpus, synth.req_and_of_ors([a, b], [c, d]), exact_scores=False)


def test_facets(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    r = synth.req_and([a, b])
    r["facets"] = [{"field": "cat"}, {"field": "tags[]", "top": 5}]
    got = check(corpus, r)
    assert got.facets and len(got.facets["cat"]) == 10 and len(got.facets["tags[]"]) == 5
    r = synth.req_single(meta.extra_probes[2])
    r["facets"] = [{"field": "tags[]", "top": 50}]
    check(corpus, r)


def test_filter(corpus):
    from veloci_amd import synth
    data, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    # identity column: filter ids are the matched TERM ids taken as anchors (search_field.rs:474-481)
    leaf = lambda t: {"search": {"path": "body", "terms": [t]}}
    r = synth.req_or([a, b, c], top=20)
    r["filter"] = {"or": {"queries": [leaf(t) for t in meta.background[:20]]}}
    check(corpus, r)


def test_boost_term(corpus):
    from veloci_amd import synth
    _, meta, _, _ = corpus
    a, b, c = meta.triples[0]
    r = synth.req_or([a, b, c], top=20)
    r["boost_term"] = [{"path": "body", "terms": [t], "boost": 3.0} for t in meta.background[:5]] + [{"path": "body", "terms": [meta.background[6]]}]
    check(corpus, r)


def test_errors(corpus):
    import veloci_amd
    _, meta, idx, ora = corpus
    req = {"search_req": {"search": {"path": "nofield", "terms": ["x"]}}}
    with pytest.raises(veloci_amd.VelociError) as ei:
        veloci_amd.search(req, idx)
    # pinned text: tests/all/tests.rs:422-436 of the reference
    assert ei.value.code == 2 and str(ei.value) == "field does not exist nofield.textindex (fst not found)"
    from oracle.binding import OracleError
    with pytest.raises(OracleError) as eo:
        ora.search_json(json.dumps(req))
    assert str(eo.value) == str(ei.value)
    with pytest.raises(veloci_amd.VelociError) as e2:
        veloci_amd.search({"top": 3}, idx)
    assert e2.value.code == 1
    with pytest.raises(veloci_amd.VelociError) as e3:
        veloci_amd.search({"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": ["x"], "options": {"explain": True}}},
                                                             {"search": {"path": "body", "terms": ["y"]}}]}}}, idx)  # explain on a part of the tree: declined
    assert e3.value.code == 4


def test_batch_equals_single(corpus):
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    _, meta, idx, ora = corpus
    reqs = []
    for tri in meta.triples:
        reqs += [synth.req_and(list(tri)), synth.req_or(list(tri)), synth.req_and_phrase_locality(list(tri)), synth.req_single(tri[0], top=3)]
    reqs += [synth.req_single(t) for t in meta.extra_probes + meta.background[:8]]
    reqs = reqs * 4
    got = veloci_amd.search_batch(reqs, idx)
    for r, g in zip(reqs, got):
        assert_same(r, g, ora.search_json(json.dumps(r)))


def test_two_shards_merge_equals_unsharded(corpus):
    """doc-range shards + packed partials + merge == the unsharded result, bit for bit."""
    import torch
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    data, meta, idx, ora = corpus
    N = data.num_anchors
    cut = N // 3
    s0 = veloci_amd.Index(data, device=0, doc_lo=0, doc_hi=cut)
    s1 = veloci_amd.Index(data, device=0, doc_lo=cut, doc_hi=N)
    a, b, c = meta.triples[0]
    reqs = [synth.req_and([a, b, c]), synth.req_or([a, b, c], top=30), synth.req_and_phrase_locality([a, b, c]), synth.req_single(meta.extra_probes[0])]
    reqs[1]["facets"] = [{"field": "cat"}, {"field": "tags[]", "top": 7}]
    reqs.append(dict(synth.req_or([a, b, c]), facets=[{"field": "tags[]", "top": None}]))  # > 1024 entries: ranked on the host from the summed histogram
    p0 = veloci_amd.PartialBatch(s0, reqs)
    p1 = veloci_amd.PartialBatch(s1, reqs)
    assert p0.nbytes == p1.nbytes
    from veloci_amd.dist import exchange_local
    g = exchange_local([p0, p1])
    res = p0.merge(g.data_ptr(), 2)
    p1.merge(None, 1)  # releases the shard's workspace
    for r, got in zip(reqs, res):
        assert_same(r, got, ora.search_json(json.dumps(r)))
    # why_found_info (why_found with select) joins the returned anchors to their texts: a shard declines it — its neighbours in the batch are answered
    with pytest.raises(veloci_amd.VelociError) as e:
        veloci_amd.search(dict(reqs[3], why_found=True, select=["body"]), s1)
    assert e.value.code == 4 and "on a shard" in str(e.value)
    # (the unsharded index goes on — to the reference's own error here: the synthetic corpus holds no parent_to_value_id store for the join)
    from oracle.binding import OracleError
    with pytest.raises(veloci_amd.VelociError) as e2:
        veloci_amd.search(dict(reqs[3], why_found=True, select=["body"]), idx)
    with pytest.raises(OracleError) as eo:
        ora.search_json(json.dumps(dict(reqs[3], why_found=True, select=["body"])))
    assert str(e2.value) == str(eo.value) == "Did not found path in indices body.textindex.parent_to_value_id"


def test_two_shards_chunked_step_with_one_exchange_equals_unsharded(corpus):
    """The N > 1 form of ShardedSearcher._one_collective, with both shards in this process: every shard scans its chunks into its partial arena
    (vq_search_batch_partial_at), the arenas' used prefixes are concatenated shard-major — what ONE all-gather produces — and every chunk
    merges out of that buffer with the prefix size as the shard stride (vq_merge_partials_flat_strided)."""
    import torch
    import veloci_amd
    from veloci_amd import synth
    from veloci_amd.dist import device_view
    from veloci_amd.search import RequestBatch
    data, meta, idx, ora = corpus
    N = data.num_anchors
    cut = N // 3
    shards = [veloci_amd.Index(data, device=0, doc_lo=0, doc_hi=cut), veloci_amd.Index(data, device=0, doc_lo=cut, doc_hi=N)]
    t = [list(x) for x in meta.triples]
    reqs = []
    for i in range(300):
        a = t[i % len(t)]
        reqs.append([synth.req_and(a), synth.req_or(a, top=25), synth.req_single(a[i % 3], top=3)][i % 3])
    batch = RequestBatch(reqs)
    subs = batch.split(3)
    pbs, prefix = [[], []], 0
    for s, shard in enumerate(shards):
        off = 0
        for c, sb in enumerate(subs):
            pb = veloci_amd.PartialBatch(shard, sb, slot=c, arena_offset=off)
            assert pb.hist_nbytes == 0
            pbs[s].append(pb)
            off += (pb.total_nbytes + 255) // 256 * 256
        assert s == 0 or off == prefix
        prefix = off
    torch.cuda.synchronize()
    gathered = torch.cat([device_view(shard.partial_arena_ptr, prefix).clone() for shard in shards])
    torch.cuda.synchronize()
    want = veloci_amd.search_batch_flat(reqs, idx, stride=25)
    n = len(reqs)
    out = (np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 25), np.uint32), np.zeros((n, 25), np.float32), np.zeros(n, np.int32))
    offset = 0
    for pb, sb in zip(pbs[0], subs):
        pb.merge_flat(gathered.data_ptr() + pb.arena_offset, 2, 25, out, offset, shard_stride=prefix)
        offset += sb.n
    for pb in pbs[1]:
        pb.merge_flat(None, 1, 25)  # releases the shard's workspaces
    for a, b in zip(out, want):
        assert np.array_equal(a, b)


def test_generic_kernel_also_matches_for_simple_queries():
    """Pure simple queries normally run on k_scan_simple; force them through the generic k_tile_scan in a
    child process (the switch is read once per process) and re-run the parity tests that cover them."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, VQ_FORCE_GENERIC="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k",
                        "single_term or test_and or test_or or nested or leaf_boost or batch_equals or two_shards"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("env", [{"VQ_UNION_OR": "1", "VQ_NO_UNION_COV": "1"}, {"VQ_NO_UNION": "1"}, {"VQ_SIMPLE_NV": "1"}, {"VQ_NO_RICH": "1"}, {"VQ_NO_RICH": "1", "VQ_NO_QUEUE": "1"}, {"VQ_FORCE_GENERIC": "1"}, {"VQ_NO_LEAF_F32": "1"},
                                 {"VQ_NO_WIDE": "1"}, {"VQ_NO_LEAF_FUSION": "1"}, {"VQ_BOOST1N_DEVICE": "1"}, {"VQ_PROBE_NO_ARR": "1"}, {"VQ_PROBE_MIN_DOCS": "40000000"}, {"VQ_NO_PROBE_OR": "1", "VQ_NO_RICH_PRUNE": "1"},
                                 pytest.param({"VQ_RING": "1"}, marks=pytest.mark.skipif(os.environ.get("VQ_TEST_RING") != "1", reason="the persistent ring kernel is opt-in (VQ_RING=1); its parity leg runs with VQ_TEST_RING=1")),
                                 {"VQ_NO_WEIGHTED_SPANS": "1", "VQ_UNION_SPAN": "512", "VQ_BATCH_HALVES": "0", "VQ_FLAT_SMALL_CHUNKS": "1", "VQ_FLAT_CHUNKS": "4"}],
                         ids=["or_on_k_scan_union_and_single_leaves_on_the_id_stream", "single_leaf_on_k_scan_simple", "k_scan_simple_8192_doc_tiles", "rich_queries_on_k_tile_scan",
                              "k_tile_scan_without_survivor_queue", "everything_on_k_tile_scan", "materialised_leaves_on_the_tile_kernels",
                              "wide_queries_on_k_tile_scan", "same_term_or_operands_not_fused", "one_to_n_boost_lists_resolved_on_the_device", "probe_operands_as_bitmap_words_only", "shipped_routing_of_shards_below_40m_docs_ands_and_ors_on_k_scan_simple", "ors_on_k_scan_simple_and_rich_queries_without_bound_pruning", "ands_on_the_persistent_ring_kernel",
                              "spans_chunks_and_host_threads_as_before_the_scheduling_changes"])
def test_alternative_kernel_routes_match(env):
    """Single leaves run on k_scan_union and ORs on k_scan_simple by default; the other assignment must give the same results.
    (tests/conftest.py moves k_scan_probe's line to 0 docs so that the parity corpora reach the headline's kernel; the leg
    `shipped_routing_of_shards_below_40m_docs` runs the same tests with the line where production has it — what every rank of a
    multi-GPU run with shards below 40 M docs executes.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    select = "single_term or test_and or test_or or nested or leaf_boost or batch_equals or two_shards or case_insensitive or random_requests_on_synthetic or phrase_and_locality or column_boosts or and_of_ors or boost_term or fuzzy or starts_with or wide_nodes or 1n_boost or random_requests_match or (reference_integration and (or_connect or minimal or simple_search or boost))"
    if all(k.startswith("VQ_PROBE") for k in env):  # the legs about the probe kernels' routing: the tests whose requests can reach those kernels
        select = "test_and or test_or or nested or leaf_boost or batch_equals or two_shards or and_probe_kernel_shapes or probe_or_kernel_shapes or random_requests_on_synthetic"
    if "VQ_NO_PROBE_OR" in env:
        select += " or probe_or_kernel_shapes"  # (the same ORs on k_scan_simple)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k", select],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_fast_div100_is_exact_for_every_f16():
    """The dense-tile path divides by 100 with a 3-instruction sequence; it must equal the correctly rounded
    f32 division for every finite f16 input (the only inputs it ever sees)."""
    import ctypes as C
    import veloci_amd
    L = veloci_amd.lib()
    L.vq_debug_div100_mismatches.restype = C.c_uint32
    import torch
    torch.cuda.init()
    assert L.vq_debug_div100_mismatches() == 0


@pytest.mark.gpu
def test_facet_select_kernels_on_crafted_histograms():
    """k_facet_select / k_facet_select_wide (facet.rs:19-23: count descending; this build orders ties by value id) against numpy on
    histograms that reach every path: short ones (one-wave kernel), long sparse ones (wide kernel, bound from the thread maxima), long
    ones whose large counters all fall to ONE thread of the wide kernel (its bound then admits more keys than the LDS buffer holds and has
    to be raised, more than once), lengths that are no multiple of 4, fewer non-zero counters than `top`, all-zero."""
    import ctypes as C
    import veloci_amd
    import torch
    torch.cuda.init()
    L = veloci_amd.lib()
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "11")))

    calls = [0]

    def check(hist, top):
        hist = np.ascontiguousarray(hist, np.uint32)
        vals, counts = np.zeros(top, np.uint32), np.zeros(top, np.uint32)
        calls[0] += 1  # (the histograms of a batch lie back to back: every alignment of the first counter)
        n = L.vq_debug_facet_select(hist.ctypes.data_as(C.c_void_p), len(hist), top, calls[0] % 4, vals.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p))
        nz = np.flatnonzero(hist)
        order = nz[np.lexsort((nz, -hist[nz].astype(np.int64)))][:top]
        assert n == len(order), (n, len(order), len(hist), top)
        assert np.array_equal(vals[:n], order.astype(np.uint32)) and np.array_equal(counts[:n], hist[order]), (len(hist), top, vals[:n][:8], order[:8])

    for nv in (1, 7, 1024, 4095, 4096, 4099, 65536, 70001):
        for top in (1, 10, 64, 65, 300):
            sparse = np.zeros(nv, np.uint32)
            k = max(1, nv // 50)
            sparse[rng.choice(nv, k, replace=False)] = rng.integers(1, 40, k)
            check(sparse, top)
            check(rng.integers(0, 5, nv), top)               # many ties
            check(np.zeros(nv, np.uint32), top)
            few = np.zeros(nv, np.uint32)
            few[rng.choice(nv, min(nv, 3), replace=False)] = 9
            check(few, top)
    # every large counter in the slots of one thread of the wide kernel (vector v belongs to thread v % 256 of its round of 1024 vectors)
    nv = 65536
    for top in (1, 10, 64):
        h = rng.integers(1, 3, nv).astype(np.uint32)
        mine = np.flatnonzero((np.arange(nv) // 4) % 256 == 5)
        h[mine] = rng.permutation(len(mine)).astype(np.uint32) + 1000
        check(h, top)
        h2 = np.full(nv, 7, np.uint32)                       # 65 536 equal counters: ties by value id, the bound climbs by value id
        check(h2, top)


# ---------------------------------------------------------------------------------------- fuzzy / prefix (K9)
@pytest.fixture(scope="module")
def words():
    import veloci_amd
    import wordcorpus
    from oracle import binding as O
    data, terms = wordcorpus.build()
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    return data, terms, idx, ora


def _part(term, **kw):
    p = {"path": "body", "terms": [term]}
    p.update(kw)
    return p


PROBES = ["majestic", "Majestic", "searhc", "saerch", "strasse", "Straße", "über", "привет", "ПРИВЕТ", "their", "thier", "weather", "niece", "東京", "a", "ab",
          "letter", "zzzzqqq", "im", "se"]


def test_fuzzy_single_term(words):
    n_hits = 0
    for t in PROBES:
        for lev in (1, 2, 3):
            for ic in (None, True, False):
                p = _part(t, levenshtein_distance=lev)
                if ic is not None:
                    p["ignore_case"] = ic
                r = check(words, {"search_req": {"search": p}, "top": 20})
                n_hits += r.num_hits
    assert n_hits > 1000


def test_starts_with(words):
    n_hits = 0
    for t in PROBES + ["m", "ma", "maj", "S", "th", "при"]:
        for lev in (None, 1):
            for ic in (None, False):
                p = _part(t, starts_with=True)
                if lev is not None:
                    p["levenshtein_distance"] = lev
                if ic is not None:
                    p["ignore_case"] = ic
                r = check(words, {"search_req": {"search": p}, "top": 20})
                n_hits += r.num_hits
    assert n_hits > 1000


def test_fuzzy_in_trees_filters_and_boosts(words):
    f = lambda t, **kw: {"search": _part(t, **kw)}
    check(words, {"search_req": {"or": {"queries": [f("majestic", levenshtein_distance=1), f("search", levenshtein_distance=2), f("there")]}}, "top": 30})
    check(words, {"search_req": {"and": {"queries": [f("the", starts_with=True), f("letter", levenshtein_distance=2)]}}, "top": 30})
    # three run-time-sized operands: the summation order follows the merged lengths (set_op.rs:388-416)
    r = check(words, {"search_req": {"and": {"queries": [f("the", starts_with=True), f("letter", levenshtein_distance=2), f("ma", starts_with=True)]}}, "top": 30})
    assert r.num_hits > 0
    r = check(words, {"search_req": {"and": {"queries": [f("s", starts_with=True), f("a", starts_with=True), f("m", starts_with=True), f("t", starts_with=True)]}}, "top": 30})
    assert r.num_hits > 0
    check(words, {"search_req": f("weather", levenshtein_distance=2), "filter": f("ma", starts_with=True), "top": 30})
    check(words, {"search_req": f("ma", starts_with=True), "boost_term": [_part("magic", levenshtein_distance=1, boost=3.0)], "top": 30})
    check(words, {"search_req": f("sea", starts_with=True, boost=1.5), "top": 15, "skip": 3})


def test_leaf_top_limits_the_expansion(words):
    for t, kw in (("s", {"starts_with": True}), ("letter", {"levenshtein_distance": 2}), ("a", {"starts_with": True})):
        for top in (1, 3, 10):
            p = _part(t, top=top, **kw)
            # the reference's final sort of the limited term hits is unstable (search_field.rs:373-376); the oracle and
            # the product both keep FST order among equal scores
            check(words, {"search_req": {"search": p}, "top": 10})


def test_fuzzy_batch_and_clamp(words):
    import veloci_amd
    from parity import assert_same
    _, terms, idx, ora = words
    reqs = []
    rng = np.random.default_rng(3)
    for i in rng.choice(len(terms), size=120, replace=False):
        t = terms[int(i)]
        reqs.append({"search_req": {"search": _part(t, levenshtein_distance=int(rng.integers(0, 9)), starts_with=bool(rng.integers(0, 2)))}, "top": 10})
    got = veloci_amd.search_batch(reqs, idx)
    for r, g in zip(reqs, got):
        assert_same(r, g, ora.search_json(json.dumps(r)))


# ------------------------------------------------------------------ the reference's integration tests, replayed (GPU half)
import refcases  # noqa: E402

REF_CASES = refcases.load()["cases"]
# requests the MI355X path declines (VQ_ERR_UNSUPPORTED -> the caller keeps its CPU path)
GPU_DECLINES = {}
_REF_IDX = {}


@pytest.mark.parametrize("case", REF_CASES, ids=[c["name"] for c in REF_CASES])
def test_reference_integration_case(case):
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    name = case["corpus"]
    if name not in _REF_IDX:
        data, docs, info = refcases.build(name)
        ora = O.OracleIndex(data.num_anchors)
        data.load_into(ora)
        _REF_IDX[name] = (veloci_amd.Index(data, device=0), ora, docs, info)
    idx, ora, docs, info = _REF_IDX[name]
    run = lambda req: veloci_amd.search(req, idx)
    if case["name"] in GPU_DECLINES:
        with pytest.raises(veloci_amd.VelociError) as e:
            run(case["request"])
        assert e.value.kind == "Unsupported" and GPU_DECLINES[case["name"]] in str(e.value)
        return
    got = refcases.check_expectations(case, docs, info, run)
    if got is not None:
        exact = not any(b.get("boost_fun") in ("Log10", "Log2") for b in case["request"].get("boost", []))
        assert_same(case["request"], got, ora.search_json(json.dumps(case["request"])), exact_scores=exact)


# ------------------------------------------------------------------ randomized differential test on the reference's own fixture corpus
def _random_request(rng, info, depth=0, extras=False):
    """Random Request over the `test_all` corpus: trees of exact / fuzzy / prefix leaves, filters, boosts, phrase pairs, locality, facets;
    extras: also one term expanded over several fields (the query generator's shape: fused leaves), regex leaves, why_found, explain."""
    text_fields = ["meanings.ger[]", "meanings.eng[]", "tags[]", "title", "ent_seq", "kanji[].text", "field1[].text", "address[].line[]"]

    def leaf():
        path = text_fields[int(rng.integers(0, len(text_fields)))]
        terms = info[path]["terms"]
        words = [t for t in terms if t.strip() and len(t) <= 20]
        t = words[int(rng.integers(0, len(words)))]
        part = {"path": path, "terms": [t]}
        r = rng.random()
        if r < 0.25:
            part["levenshtein_distance"] = int(rng.integers(1, 3))
        elif r < 0.4:
            part["terms"] = [t[:max(1, len(t) // 2)]]
            part["starts_with"] = True
        if rng.random() < 0.2:
            part["ignore_case"] = bool(rng.integers(0, 2))
        if rng.random() < 0.2:
            part["boost"] = float(rng.choice([0.5, 2.0, 3.5]))
        if rng.random() < 0.1:
            part["top"] = int(rng.integers(1, 4))
        if extras and rng.random() < 0.06 and len(t) >= 3 and t.isalnum():
            part = {"path": path, "terms": [".*" + t[1:3] + ".*"], "is_regex": True}
        return {"search": part}

    def expansion():
        """one term over 2-6 fields, as query_parser_to_veloci_request.rs:84-109 writes it"""
        base = leaf()["search"]
        fields = [text_fields[int(i)] for i in rng.choice(len(text_fields), size=int(rng.integers(2, 7)), replace=False)]
        return [{"search": dict(base, path=f)} for f in fields]

    def tree(d):
        if d >= 2 or rng.random() < 0.4:
            return leaf()
        kind = "and" if rng.random() < 0.4 else "or"
        wide = d >= 1 and rng.random() < 0.1  # (one leaf per term and field, as the query generator writes them)
        queries = [tree(d + 1) for _ in range(int(rng.integers(5, 8) if wide else rng.integers(2, 4)))]
        if extras and kind == "or" and rng.random() < 0.35:
            for _ in range(int(rng.integers(1, 3))):
                queries[int(rng.integers(0, len(queries) + 1)):0] = expansion()
        return {kind: {"queries": queries}}

    req = {"search_req": tree(0), "top": int(rng.choice([1, 3, 10, 50]))}
    if rng.random() < 0.2:
        req["skip"] = int(rng.integers(0, 3))
    if rng.random() < 0.04:  # deep paging (beyond one scan's 1024 ranked hits)
        req["top"], req["skip"] = int(rng.choice([5, 1500])), int(rng.choice([0, 1030]))
    if rng.random() < 0.25:
        req["filter"] = tree(1)
    if rng.random() < 0.3:
        req["boost"] = [{"path": "commonness", "boost_fun": str(rng.choice(["Multiply", "Add", "Replace", "Log10", "Log2"])), "param": float(rng.choice([0.0, 1.0, 2.0]))}]
        if rng.random() < 0.3:
            req["boost"].append({"path": str(rng.choice(["kanji[].commonness", "field1[].rank", "kana[].commonness"])), "boost_fun": "Multiply", "param": 1.0})
        if rng.random() < 0.2:
            req["boost"][0]["skip_when_score"] = [0.0, 10.0]
        if rng.random() < 0.2:
            req["boost"][0]["expression"] = str(rng.choice(["$SCORE * 2", "10 / $SCORE", "$SCORE + 1.5", "3 - $SCORE"]))
    if rng.random() < 0.25:
        a, b = leaf()["search"], leaf()["search"]
        b["path"] = a["path"]
        b["terms"] = [info[a["path"]]["terms"][int(rng.integers(0, len(info[a["path"]]["terms"])))]]
        for p in (a, b):
            for key in ("levenshtein_distance", "starts_with", "boost"):
                p.pop(key, None)
        req["phrase_boosts"] = [{"search1": a, "search2": b}]
    if rng.random() < 0.2:
        bt = leaf()["search"]
        bt.pop("boost", None)
        req["boost_term"] = [dict(bt, boost=float(rng.choice([2.0, 5.0])))]
    if rng.random() < 0.3:
        req["text_locality"] = True
    if rng.random() < 0.3:
        req["facets"] = [{"field": str(rng.choice(["tags[]", "commonness", "meanings.eng[]"]))}]
    if extras and rng.random() < 0.15:
        req["why_found"] = True
    if extras and rng.random() < 0.2 and "phrase_boosts" not in req and not any("[]" in b["path"] for b in req.get("boost", [])):
        req["explain"] = True  # (with phrase boosts or a 1:n boost: declined)
    return req


def test_random_requests_on_reference_corpus_match_the_oracle():
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    data, docs, info = refcases.build("test_all")
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "20241003")))
    ran = declined = 0
    for i in range(600):
        req = _random_request(rng, info, extras=i % 2 == 1)
        js = json.dumps(req)
        try:
            want = ora.search_json(js)
        except O.OracleError as e:
            with pytest.raises(veloci_amd.VelociError) as g:
                veloci_amd.search(req, idx)
            assert str(g.value) == str(e), js
            continue
        try:
            got = veloci_amd.search(req, idx)
        except veloci_amd.VelociError as e:
            # the one documented decline the generator can reach: which values the reference applies follows the leaf's
            # hits around the anchor (resolved by a range pre-pass) — but under a Set filter those hits are the filtered ones
            assert e.kind == "Unsupported" and "several boosted values on one anchor, under a filter" in str(e), (str(e), js)
            declined += 1
            continue
        exact = not any(b.get("boost_fun") in ("Log10", "Log2") for b in req.get("boost", []))
        assert_same(req, got, want, exact_scores=exact)
        assert {k: sorted(v) for k, v in got.why_found_terms.items()} == {k: sorted(v) for k, v in want.why_found_terms.items()}, js
        if exact:
            assert got.explain_json == (want.explain_json if req.get("explain") else "null"), js
        ran += 1
    assert declined <= 6 and ran >= 585, (ran, declined)


def test_1n_boost_with_several_values_per_anchor_follows_the_reference_walk():
    """boost.rs:255-281 merges the (anchor, value) list of a 1:n boost into the leaf's hits with one look-ahead entry: an anchor with
    several boosted values gets the first or all of them depending on the hits around it.  Random 1:n objects, leaves that match
    several of a doc's entries (prefix / fuzzy / exact), every boost function — all docs compared with the oracle."""
    import veloci_amd
    from veloci_amd import mini_indexer
    from oracle import binding as O
    from parity import assert_same
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "515")))
    words = ["alpha", "alpine", "alps", "alto", "beta", "bet", "gamma"]
    docs = []
    for d in range(500):
        items = []
        for _ in range(int(rng.choice([0, 0, 1, 2, 3, 4]))):
            it = {"text": str(rng.choice(words))}
            if rng.random() < 0.8:
                it["rank"] = int(rng.integers(1, 9))
            items.append(it)
        doc = {"title": str(rng.choice(words))}
        if items:
            doc["items"] = items
        docs.append(doc)
    indices = {"*GLOBAL*": {"features": ["All"]}, "items[].text": {}, "items[].rank": {"boost": {"boost_type": "f32"}}, "title": {}}
    data, info = mini_indexer.build_index(docs, indices)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    reqs, wants = [], []
    for i in range(120):
        leaf = {"path": "items[].text", "terms": [str(rng.choice(["al", "alp", "alpha", "alps", "bet", "beta", "a"]))]}
        kind = rng.random()
        if kind < 0.5:
            leaf["starts_with"] = True
        elif kind < 0.75:
            leaf["levenshtein_distance"] = int(rng.integers(1, 3))
        tree = {"search": leaf}
        if rng.random() < 0.3:
            tree = {str(rng.choice(["and", "or"])): {"queries": [tree, {"search": {"path": "title", "terms": [str(rng.choice(words))]}}]}}
        req = {"search_req": tree, "top": 600,
               "boost": [{"path": "items[].rank", "boost_fun": str(rng.choice(["Multiply", "Add", "Replace"])), "param": float(rng.choice([0.0, 1.0]))}]}
        if rng.random() < 0.2:
            req["boost"][0]["expression"] = "$SCORE * 2"
        want = ora.search_json(json.dumps(req))
        got = veloci_amd.search(req, idx)
        assert_same(req, got, want, exact_scores=True)
        reqs.append(req)
        wants.append(want)
    # the same requests as one batch over three doc-range shards: the range counts are summed over the shards
    for req, g, want in zip(reqs, _search_batch_over_shards(data, reqs, 3), wants):
        assert not isinstance(g, Exception), (str(g), json.dumps(req))
        assert_same(req, g, want, exact_scores=True)


def _jmdict_request(term, lev):
    """Request of the reference's own benchmark, benches/bench_jmdict.rs:115-235 (`get_request(term, levenshtein_distance)`): an OR
    over five field leaves, each with its own anchor-level and 1:n boosts (BASELINE.json configs[0])."""
    def leaf(path, boosts, starts_with):
        part = {"terms": [term], "path": path, "levenshtein_distance": lev, "options": {"boost": boosts}}
        if starts_with:
            part["starts_with"] = True
        return {"search": part}
    common = lambda p: {"path": "commonness", "boost_fun": "Log10", "param": p}
    return {"search_req": {"or": {"queries": [
        leaf("kanji[].text", [common(1), {"path": "kanji[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("kana[].text", [common(1), {"path": "kana[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("kana[].text", [common(1), {"path": "kana[].commonness", "boost_fun": "Log10", "param": 1}], True),
        leaf("meanings.ger[].text", [common(0), {"path": "meanings.ger[].rank", "expression": "10 / $SCORE"}], False),
        leaf("meanings.eng[]", [common(1)], False),
    ], "options": {"top": 10, "skip": 0}}}}


def test_bench_jmdict_request_shape_matches_the_oracle():
    """BASELINE.json configs[0]: the request shape of the reference's bench_jmdict on a JMdict-like corpus with the index
    configuration of veloci_bins/src/bin/create_test_index.rs:33-69 (jmdict.json itself is a git-lfs pointer in the reference)."""
    import veloci_amd
    from veloci_amd import mini_indexer
    from oracle import binding as O
    from parity import assert_same
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "166600")))
    kanji = ["意慾", "意欲", "慾意", "威容", "偉容", "意地", "意図", "異様"]
    kana = ["いよく", "いよう", "いじ", "いと", "いよ", "よく"]
    eng = ["will", "desire", "ambition", "dignity", "majestic appearance", "will power", "intention", "strange", "well", "wish"]
    ger = ["Wille", "Wunsch", "Begehren", "Ehrgeiz", "majestätischer Anblick", "Absicht", "Wollen", "Willenskraft"]
    docs = []
    for d in range(2000):
        doc = {"commonness": int(rng.choice([0, 5, 20, 350, 3000])), "ent_seq": str(1000000 + d), "pos": [str(rng.choice(["n", "v1", "adj-i"]))]}
        doc["kanji"] = [{"text": str(rng.choice(kanji)), "commonness": int(rng.choice([0, 3, 40, 500]))} for _ in range(int(rng.integers(0, 3)))]
        doc["kana"] = [{"text": str(rng.choice(kana)), "romaji": "iyoku", "commonness": int(rng.choice([0, 7, 60]))} for _ in range(int(rng.integers(1, 3)))]
        doc["meanings"] = {"eng": [str(rng.choice(eng)) for _ in range(int(rng.integers(1, 4)))],
                           "ger": [{"text": str(rng.choice(ger)), "rank": int(rng.integers(1, 6))} for _ in range(int(rng.integers(0, 3)))]}
        if not doc["kanji"]:
            del doc["kanji"]
        if not doc["meanings"]["ger"]:
            del doc["meanings"]["ger"]
        docs.append(doc)
    indices = {"commonness": {"boost": {"boost_type": "f32"}}, "meanings.ger[].rank": {"boost": {"boost_type": "f32"}},
               "kanji[].commonness": {"boost": {"boost_type": "f32"}}, "kana[].commonness": {"boost": {"boost_type": "f32"}},
               "kanji[].text": {"fulltext": {"tokenize": False}}, "kana[].text": {"fulltext": {"tokenize": False}},
               "kana[].romaji": {"fulltext": {"tokenize": True}}, "meanings.ger[].text": {"fulltext": {"tokenize": True}},
               "meanings.eng[]": {"fulltext": {"tokenize": True}}, "pos": {"fulltext": {"tokenize": False}}}
    data, info = mini_indexer.build_index(docs, indices)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    ran = 0
    for term in ["意慾", "意", "いよ", "いよく", "will", "Wille", "majestic", "desire", "Absicht", "well"]:
        for lev in (0, 1):
            req = _jmdict_request(term, lev)
            js = json.dumps(req)
            try:
                want = ora.search_json(js)
            except O.OracleError as e:
                with pytest.raises(veloci_amd.VelociError) as g:
                    veloci_amd.search(req, idx)
                assert str(g.value) == str(e), js
                continue
            got = veloci_amd.search(req, idx)
            assert_same(req, got, want, exact_scores=False)  # Log10 boosts: device log vs glibc, 1e-5
            ran += 1
    assert ran >= 16, ran


def test_text_locality_on_multi_valued_text_fields_device_prepass():
    """A9 / K7 on the device: text locality of fields whose text ids are NOT anchors (boost.rs:34-87) — 1:n text fields whose texts repeat
    across documents (one text id -> many anchors), terms matching many tokens (prefix / fuzzy: long token -> text rows, counts c > 2), two
    such fields in one request (the smallest boost per anchor wins, boost.rs:25), single / batched / two doc-range shards; 30 000 documents,
    far beyond what the old host path's tests covered."""
    import veloci_amd
    from veloci_amd import mini_indexer
    from oracle import binding as O
    from parity import assert_same
    rng = np.random.default_rng(9)
    words = ["alpha", "alpine", "alps", "beta", "betal", "gamma", "gamut", "delta", "deltoid", "omega", "omen", "river", "rival", "stone", "story", "storm"]
    phrases = [" ".join(rng.choice(words, int(rng.integers(2, 5)))) for _ in range(400)]  # texts repeat across documents
    notes = [" ".join(rng.choice(words, int(rng.integers(2, 6)))) for _ in range(3000)]
    docs = []
    for d in range(30_000):
        doc = {"title": str(rng.choice(phrases)), "lines": [{"text": str(rng.choice(phrases))} for _ in range(int(rng.integers(1, 4)))],
               "notes": [str(rng.choice(notes)) for _ in range(int(rng.integers(0, 3)))]}
        if not doc["notes"]:
            del doc["notes"]
        docs.append(doc)
    indices = {"title": {"fulltext": {"tokenize": True}}, "lines[].text": {"fulltext": {"tokenize": True}}, "notes[]": {"fulltext": {"tokenize": True}}}
    data, info = mini_indexer.build_index(docs, indices)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    leaf = lambda path, t, **kw: {"search": dict({"path": path, "terms": [t]}, **kw)}
    reqs = []
    for path in ("lines[].text", "notes[]", "title"):
        for a, b in (("alpha", "beta"), ("gamma", "delta"), ("river", "stone"), ("omega", "storm")):
            reqs.append({"search_req": {"or": {"queries": [leaf(path, a), leaf(path, b)]}}, "text_locality": True, "top": 20})
            reqs.append({"search_req": {"and": {"queries": [leaf(path, a), leaf(path, b)]}}, "text_locality": True, "top": 20})
        reqs.append({"search_req": {"or": {"queries": [leaf(path, "al", starts_with=True), leaf(path, "st", starts_with=True), leaf(path, "delta", levenshtein_distance=1)]}},
                     "text_locality": True, "top": 30})
    reqs.append({"search_req": {"or": {"queries": [leaf("lines[].text", "alpha"), leaf("lines[].text", "beta"), leaf("notes[]", "alpha"), leaf("notes[]", "beta"),
                                                   leaf("title", "alpha"), leaf("title", "beta")]}}, "text_locality": True, "top": 50})
    reqs.append({"search_req": {"or": {"queries": [leaf("lines[].text", "gamma"), leaf("lines[].text", "gamma"), leaf("lines[].text", "omen")]}}, "text_locality": True})
    wants = [ora.search_json(json.dumps(r)) for r in reqs]
    assert sum(w.num_hits > 100 for w in wants) > len(wants) // 2
    for r, w in zip(reqs, wants):
        assert_same(r, veloci_amd.search(r, idx), w)
    for r, g, w in zip(reqs * 4, veloci_amd.search_batch(reqs * 4, idx), wants * 4):
        assert_same(r, g, w)
    for r, g, w in zip(reqs, _search_batch_over_shards(data, reqs, 2), wants):
        assert not isinstance(g, Exception), (str(g), json.dumps(r))
        assert_same(r, g, w)


def test_highlight_matches_the_reference_and_the_oracle():
    """search_field::highlight (search_field.rs:233-245, highlight_field.rs:187-272) through the C ABI — the reference's own three
    assertions (tests/all/tests.rs:1009-1085), then product == oracle (snippets, score bits, text ids, error texts) on the wider request set,
    whose prefix / fuzzy parts take their match sets from k_dict_scan."""
    import veloci_amd
    from oracle import binding as O
    import test_reference_integration as T
    fx = T._load_suggest_regex()
    data, docs, info = T.build_fixture_corpus(fx, "test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    idx = veloci_amd.Index(data, device=0)
    for case in fx["highlight"]:
        assert [t for t, _, _ in veloci_amd.highlight(case["request"], idx)] == case["expect_texts"], case["name"]
    answered = errors = 0
    for part in T.highlight_parts():
        js = json.dumps(part)
        try:
            want = ora.highlight_json(js)
        except O.OracleError as e:
            with pytest.raises(veloci_amd.VelociError) as g:
                veloci_amd.highlight(part, idx)
            assert str(g.value) == str(e), js
            errors += 1
            continue
        got = veloci_amd.highlight(part, idx)
        assert [(t, np.float32(s).view(np.uint32), i) for t, s, i in got] == [(t, np.float32(s).view(np.uint32), i) for t, s, i in want], js
        answered += 1
    assert answered > 150 and errors > 10, (answered, errors)


def test_suggest_regex_and_why_found_terms_match_the_reference_and_the_oracle():
    """SURVEY.md §8f-5 / f-4 (dictionary side): suggest (search_field.rs:194-231), regex leaves (:72-83) and `why_found_terms` (search.rs:186)
    through the C ABI — the reference's own assertions (tests/golden/reference_suggest_regex.json), then product == oracle on a wider set."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    import test_reference_integration as T
    fx = T._load_suggest_regex()
    built = {}

    def corpus_of(name, tv=None):
        key = (name, json.dumps(tv))
        if key not in built:
            data, docs, info = T.build_fixture_corpus(fx, name, tv)
            ora = O.OracleIndex(data.num_anchors)
            data.load_into(ora)
            built[key] = (veloci_amd.Index(data, device=0), ora, docs, info)
        return built[key]

    for case in fx["suggest"]:
        idx, ora, docs, info = corpus_of(case["corpus"], case.get("token_values"))
        got = veloci_amd.suggest(case["request"], idx)
        T.check_suggest_case(case, got)
        want = ora.suggest_json(json.dumps(case["request"]))
        assert [(t, np.float32(s).view(np.uint32)) for t, s, _ in got] == [(t, np.float32(s).view(np.uint32)) for t, s, _ in want], case["name"]
    for case in fx["term_lookup"]:
        idx, ora, docs, info = corpus_of(case["corpus"])
        assert sorted(t for t, _, _ in veloci_amd.suggest(case["request"], idx)) == case["expect_terms_sorted_lowercase"], case["name"]
    for case in fx["regex"]:
        idx, ora, docs, info = corpus_of(case["corpus"])
        res = veloci_amd.search(case["request"], idx)
        assert len(res.ids) == case["expect_len"], (case["name"], res.ids)
        if "expect_doc0" in case:
            assert docs[int(res.ids[0])][case["expect_doc0"][0]] == case["expect_doc0"][1], case["name"]
        assert_same(case["request"], res, ora.search_json(json.dumps(case["request"])))
    # wider: suggest parts of every kind, regex leaves inside trees, why_found_terms of trees
    idx, ora, docs, info = corpus_of("test_all")
    parts = []
    for path in ("meanings.ger[]", "meanings.eng[]", "title", "kanji[].text"):
        for term, kw in (("will", {"starts_with": True}), ("majes", {"starts_with": True, "top": 3}), ("wille", {"levenshtein_distance": 1}), ("Begeisterung", {"levenshtein_distance": 2, "skip": 1, "top": 2}),
                         (".*e.*", {"is_regex": True}), ("w.*", {"is_regex": True, "ignore_case": False}), ("der", {}), ("majestät", {"levenshtein_distance": 1, "boost": 2.5})):
            parts.append(dict({"terms": [term], "path": path}, **kw))
    for part in parts:
        js = json.dumps(part)
        try:
            want = ora.suggest_json(js)
        except O.OracleError as e:
            with pytest.raises(veloci_amd.VelociError) as g:
                veloci_amd.suggest(part, idx)
            assert str(g.value) == str(e), js
            continue
        got = veloci_amd.suggest(part, idx)
        assert sorted((t, np.float32(s).view(np.uint32)) for t, s, _ in got) == sorted((t, np.float32(s).view(np.uint32)) for t, s, _ in want), js
        assert [np.float32(s).view(np.uint32) for _, s, _ in got] == [np.float32(s).view(np.uint32) for _, s, _ in want], js
    multi = {"suggest": parts[:6], "top": 7, "skip": 1}
    assert [(t, s) for t, s, _ in veloci_amd.suggest(multi, idx)] == [(t, s) for t, s, _ in ora.suggest_json(json.dumps(multi))]
    leaf = lambda **kw: {"search": kw}
    reqs = [
        {"search_req": leaf(terms=[".*wil.*"], path="meanings.ger[]", is_regex=True), "why_found": True},
        {"search_req": {"or": {"queries": [leaf(terms=["will"], path="meanings.ger[]", levenshtein_distance=1), leaf(terms=["will"], path="meanings.eng[]"),
                                           leaf(terms=[".*ung.*"], path="meanings.ger[]", is_regex=True)]}}, "why_found": True, "top": 20},
        {"search_req": {"and": {"queries": [leaf(terms=["m.*"], path="meanings.ger[]", is_regex=True, ignore_case=False), leaf(terms=["majes"], path="meanings.ger[]", starts_with=True)]}},
         "why_found": True, "filter": leaf(terms=["der"], path="meanings.ger[]")},
        {"search_req": {"or": {"queries": [leaf(terms=["will"], path="meanings.eng[]"), leaf(terms=["will"], path="meanings.eng[]")]}}, "why_found": True},
        {"search_req": leaf(terms=["[0-9]+.*"], path="meanings.ger[]", is_regex=True)},
    ]
    for req in reqs:
        want = ora.search_json(json.dumps(req))
        got = veloci_amd.search(req, idx)
        assert_same(req, got, want)
        assert {k: sorted(v) for k, v in got.why_found_terms.items()} == {k: sorted(v) for k, v in want.why_found_terms.items()}, json.dumps(req)
    for req, got in zip(reqs, veloci_amd.search_batch(reqs, idx)):
        assert_same(req, got, ora.search_json(json.dumps(req)))
    # `select` alone changes nothing in search::search (reading the fields is to_documents' work, search.rs:63-103)
    assert_same(reqs[0], veloci_amd.search(dict(reqs[0], select=["title"]), idx), ora.search_json(json.dumps(reqs[0])))


def test_why_found_with_select_matches_the_reference_and_the_oracle():
    """`why_found` together with `select` (search.rs:220-224 -> SearchResult::why_found_info, search/why_found.rs:11-50) through the C ABI: the
    reference's own assertions (tests/golden/reference_why_found.json), then product == oracle on a wider set — singly and as one batch — and the
    entry points without a place for the map decline."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    from veloci_amd import mini_indexer
    import refcases
    import test_reference_integration as T
    fx = T._load_why_found()
    built = {}
    for case in fx["cases"]:
        if case["corpus"] not in built:
            c = fx["corpora"][case["corpus"]]
            docs = refcases.corpus_docs(c)
            data, info = mini_indexer.build_index(docs, c["indices"])
            ora = O.OracleIndex(data.num_anchors)
            data.load_into(ora)
            built[case["corpus"]] = (veloci_amd.Index(data, device=0), ora, docs)
        idx, ora, docs = built[case["corpus"]]
        got = veloci_amd.search(case["request"], idx)
        assert len(got.ids) >= 1, case["name"]
        for k, v in case.get("expect_doc", {}).items():
            assert docs[int(got.ids[0])][k] == v, case["name"]
        assert got.why_found_info.get(int(got.ids[0]), {}) == case["expect_why_found"], (case["name"], got.why_found_info)
        want = ora.search_json(json.dumps(case["request"]))
        assert_same(case["request"], got, want)
        assert got.why_found_info == want.why_found_info, case["name"]

    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    idx = veloci_amd.Index(data, device=0)
    reqs = T.why_found_select_requests()
    answered, with_entries = [], 0
    for req in reqs:
        try:
            want = ora.search_json(json.dumps(req))
        except O.OracleError as eo:
            with pytest.raises(veloci_amd.VelociError) as ev:
                veloci_amd.search(req, idx)
            assert str(ev.value) == str(eo), req
            continue
        got = veloci_amd.search(req, idx)
        assert_same(req, got, want)
        if req.get("why_found") and "select" in req:
            assert got.why_found_info == want.why_found_info, (json.dumps(req), got.why_found_info, want.why_found_info)
            with_entries += sum(len(t) for f in got.why_found_info.values() for t in f.values())
        else:
            assert got.why_found_info is None and want.why_found_info == {}, req
        answered.append((req, want))
    assert len(answered) > 40 and with_entries > 40, (len(answered), with_entries)
    for (req, want), got in zip(answered, veloci_amd.search_batch([r for r, _ in answered], idx)):
        assert_same(req, got, want)
        assert (got.why_found_info or {}) == want.why_found_info, json.dumps(req)
    # the flat entry point has no place for the map: such a request is declined there, its neighbours are answered
    asked = [r for r, _ in answered if r.get("why_found") and "select" in r][:3]
    plain = [r for r, _ in answered if not r.get("why_found")][:1]
    status = veloci_amd.search_batch_flat(asked + plain, idx, stride=20)[4]
    assert len(asked) == 3 and list(status) == [4, 4, 4, 0], status


def test_random_requests_in_batches_match_the_oracle():
    """The same generator through vq_search_batch: dictionary scans, union jobs and count pre-passes of many requests share one batch."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    data, docs, info = refcases.build("test_all")
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "77")))
    for _ in range(4):
        reqs = [_random_request(rng, info) for _ in range(300)]
        got = veloci_amd.search_batch(reqs, idx, raise_on_error=False)
        for req, g in zip(reqs, got):
            js = json.dumps(req)
            try:
                want = ora.search_json(js)
            except O.OracleError as e:
                assert isinstance(g, veloci_amd.VelociError) and g.code == e.code, js  # (the batch entry point reports codes, not texts)
                continue
            if isinstance(g, veloci_amd.VelociError) and g.kind == "Unsupported" and any("[]" in b["path"] for b in req.get("boost", [])) and "filter" in req:
                continue  # 1:n boost with several boosted values on one anchor, under a filter (see the single-request test)
            assert not isinstance(g, Exception), (str(g), js)
            exact = not any(b.get("boost_fun") in ("Log10", "Log2") for b in req.get("boost", []))
            assert_same(req, g, want, exact_scores=exact)


@pytest.fixture(scope="module")
def big_corpus():
    import veloci_amd
    from veloci_amd import synth
    from oracle import binding as O
    spec = synth.SynthSpec(num_docs=4_000_000, num_terms=3000, triples=3, extra_probe_dfs=(5000, 400_000), background_terms=30)
    data, meta = synth.generate(spec)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    return data, meta, idx, ora


def test_random_requests_on_a_4m_doc_corpus_match_the_oracle(big_corpus):
    """The same generator on 4 M docs: many spans per query, sequential tiles, the shared threshold word, pruning tables."""
    _random_synthetic(big_corpus, n_requests=70, seed=int(os.environ.get("VQ_TEST_SEED", "4004")))


def test_single_requests_split_into_many_spans_merge_exactly(big_corpus):
    """A batch of one is split into many more spans than a full batch (latency path): the span merge's bound filter (top <= 64) and
    its general rounds (top > 64), AND / OR / single leaf, against the oracle."""
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    data, meta, idx, ora = big_corpus
    t = list(meta.triples[0])
    for top in (1, 10, 64, 65, 300):
        for skip in (0, 7):
            for req in (synth.req_and(t, top=top), synth.req_or(t, top=top), synth.req_single(t[0], top=top),
                        synth.req_and_of_ors([t[0], t[1]], [t[2], meta.triples[1][0]], top=top)):
                req = dict(req, skip=skip)
                got = veloci_amd.search(req, idx)
                want = ora.search_json(json.dumps(req))
                assert_same(req, got, want, exact_scores="boost" not in req)  # (Log10: device log vs glibc, 1e-5)


def test_and_probe_kernel_shapes(corpus, big_corpus):
    """k_scan_probe (ANDs of one id-list cover and 1-3 bitmap operands): every operand order (the summation order follows the list
    lengths, set_op.rs:393), top from 1 to beyond one candidate buffer, skip, deep pages (key_upper), leaf boosts (term scores that differ
    per operand, a negative one: nothing may be pruned then), single requests (many spans) and batches, against the oracle."""
    import itertools
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    for cp in (corpus, big_corpus):
        data, meta, idx, ora = cp
        a, b = list(meta.triples[0]), list(meta.triples[1])
        reqs = []
        for terms in itertools.permutations(a):
            reqs.append(synth.req_and(list(terms)))
        reqs += [synth.req_and([a[0], a[2]]), synth.req_and([a[2], a[1]], top=1), synth.req_and([a[0], a[1], b[0], a[2]], top=20), synth.req_and([b[2], a[0], b[0], a[1]], top=3),
                 synth.req_and([a[0], b[1], b[2]], top=100), dict(synth.req_and(a, top=700), skip=300), dict(synth.req_and(a, top=5), skip=2000), synth.req_and(a, top=1500),
                 dict(synth.req_and([a[1], a[2]], top=10), skip=10**7)]
        leaf = lambda t, boost=None: {"search": dict({"path": "body", "terms": [t]}, **({"boost": boost} if boost is not None else {}))}
        reqs += [{"search_req": {"and": {"queries": [leaf(a[0], 2.5), leaf(a[1]), leaf(a[2], 0.5)]}}, "top": 10},
                 {"search_req": {"and": {"queries": [leaf(a[0], -1.0), leaf(a[1]), leaf(a[2])]}}, "top": 10},
                 {"search_req": {"and": {"queries": [leaf(a[0]), leaf(a[1]), leaf(a[2], -2.0)]}}, "top": 10},
                 {"search_req": {"and": {"queries": [leaf(a[0], 0.0), leaf(a[2], 3.0)]}}, "top": 10}]
        wants = [ora.search_json(json.dumps(r)) for r in reqs]
        for r, w in zip(reqs, wants):
            assert_same(r, veloci_amd.search(r, idx), w)
        for r, g, w in zip(reqs * 3, veloci_amd.search_batch(reqs * 3, idx), wants * 3):
            assert_same(r, g, w)
        for r, g, w in zip(reqs, _search_batch_over_shards(data, reqs, 2), wants):
            assert not isinstance(g, Exception), (str(g), json.dumps(r))
            assert_same(r, g, w)


def test_probe_or_kernel_shapes_and_reruns(corpus, big_corpus):
    """k_scan_probe_or (ORs of 2-3 leaves with a term slot each — the 4-leaf requests below run on k_scan_simple —, the sparsest operand streamed as the cover, the others read as bitmap words): it
    counts the union from the words and ranks only the docs that hold the cover; finish_batch confirms the result by its k-th key or runs the
    request again on the exact kernels (set_op.rs:87-220).  Every operand order, top from 1 to beyond the candidate buffer, skip, deep pages,
    leaf boosts — among them ones that push the cover's docs BELOW the others' (the short cut must be refused: vq_index_speculative_reruns grows) —,
    a cover with fewer hits than `top`, single requests and batches, against the oracle."""
    import itertools
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    L = veloci_amd.lib()
    for cp in (corpus, big_corpus):
        data, meta, idx, ora = cp
        a, b = list(meta.triples[0]), list(meta.triples[1])
        leaf = lambda t, boost=None: {"search": dict({"path": "body", "terms": [t]}, **({"boost": boost} if boost is not None else {}))}
        orq = lambda leaves, **kw: dict({"search_req": {"or": {"queries": leaves}}, "top": 10}, **kw)
        reqs = [synth.req_or(list(terms)) for terms in itertools.permutations(a)]
        reqs += [synth.req_or([a[0], a[2]]), synth.req_or([a[2], a[1]], top=1), synth.req_or([a[0], a[1], b[0], a[2]], top=20), synth.req_or([b[2], a[0], b[0], a[1]], top=3),
                 synth.req_or([a[0], b[1], b[2]], top=100), dict(synth.req_or(a, top=700), skip=300), dict(synth.req_or(a, top=5), skip=2000), synth.req_or(a, top=1500),
                 dict(synth.req_or([a[1], a[2]], top=10), skip=10**7)]
        wants = [ora.search_json(json.dumps(r)) for r in reqs]
        before = int(L.vq_index_speculative_reruns(idx.h))
        for r, w in zip(reqs[:11], wants[:11]):  # (windows of at most 20 hits: the planted overlap puts them into all the lists — nothing to run again)
            assert_same(r, veloci_amd.search(r, idx), w)
        plain_reruns = int(L.vq_index_speculative_reruns(idx.h)) - before
        for r, w in zip(reqs[11:], wants[11:]):
            assert_same(r, veloci_amd.search(r, idx), w)
        for r, g, w in zip(reqs * 3, veloci_amd.search_batch(reqs * 3, idx), wants * 3):
            assert_same(r, g, w)
        # the rarest term's postings made worthless / the other terms' boosted: the best hits no longer hold the cover — and a window
        # (top + skip) that reaches beyond the cover's hits
        tricky = [orq([leaf(a[0], 5.0), leaf(a[1], 4.0), leaf(a[2], 0.001)]), orq([leaf(a[0]), leaf(a[1]), leaf(a[2], 0.0)]), orq([leaf(a[0], 9.0), leaf(a[2], 0.01)], top=50),
                  orq([leaf(a[0]), leaf(a[1], -1.0), leaf(a[2])]), orq([leaf(a[0]), leaf(a[1]), leaf(a[2], -2.0)]), dict(synth.req_or(a, top=400), skip=10**5 * (4 if cp is big_corpus else 1) // 10)]
        before = int(L.vq_index_speculative_reruns(idx.h))
        twants = [ora.search_json(json.dumps(r)) for r in tricky]
        for r, g, w in zip(tricky, veloci_amd.search_batch(tricky, idx), twants):
            assert_same(r, g, w)
        for r, w in zip(tricky, twants):
            assert_same(r, veloci_amd.search(r, idx), w)
        route_on = os.environ.get("VQ_PROBE_MIN_DOCS") == "0" and not any(os.environ.get(k) for k in ("VQ_FORCE_GENERIC", "VQ_NO_PROBE", "VQ_NO_PROBE_OR"))
        if route_on:  # (the legs of test_alternative_kernel_routes_match run this test on other kernels: nothing is speculative there)
            assert int(L.vq_index_speculative_reruns(idx.h)) - before >= 4, "requests whose best hits lack the cover must be run again on the exact kernels"
            assert plain_reruns == 0, plain_reruns  # (the planted overlap puts the best hits into all three lists)


def test_probe_or_random_requests_with_leaf_boosts(big_corpus):
    """240 random ORs of 2-3 terms (every df of the corpus, repeated terms, leaf boosts from -2 to 20 — zero, tiny and negative ones among them —, top 1-300,
    skips up to beyond the hits) in batches: whatever k_scan_probe_or confirms and whatever it has to hand back to the exact kernels must equal the oracle
    (set_op.rs:87-220); some of both must occur."""
    import veloci_amd
    from parity import assert_same
    data, meta, idx, ora = big_corpus
    rng = np.random.default_rng(int(os.environ.get("VQ_TEST_SEED", "77")))
    pool = [t for tri in meta.triples for t in tri] + list(meta.extra_probes) + list(meta.background[:6])
    boosts = [None, None, None, 1.0, 2.5, 0.5, 0.01, 0.0, 20.0, -1.0, 1e-7]
    reqs = []
    for _ in range(240):
        n = int(rng.integers(2, 4))
        leaves = []
        for t in rng.choice(len(pool), n, replace=bool(rng.random() < 0.1)):
            b = boosts[int(rng.integers(0, len(boosts)))]
            leaves.append({"search": dict({"path": "body", "terms": [pool[int(t)]]}, **({"boost": b} if b is not None else {}))})
        r = {"search_req": {"or": {"queries": leaves}}, "top": int(rng.choice([1, 3, 10, 10, 10, 40, 300]))}
        if rng.random() < 0.2:
            r["skip"] = int(rng.choice([1, 7, 100, 5000, 10**7]))
        reqs.append(r)
    before = idx.speculative_reruns
    for lo in range(0, len(reqs), 80):
        part = reqs[lo:lo + 80]
        for r, g in zip(part, veloci_amd.search_batch(part, idx)):
            assert_same(r, g, ora.search_json(json.dumps(r)))
    again = idx.speculative_reruns - before
    if os.environ.get("VQ_PROBE_MIN_DOCS") == "0" and not any(os.environ.get(k) for k in ("VQ_FORCE_GENERIC", "VQ_NO_PROBE", "VQ_NO_PROBE_OR")):
        assert 0 < again < len(reqs), again


def test_full_size_index_matches_the_oracle():
    """BASELINE.json's full size on one GPU: the bench's 100 M-doc index (one probe triple, all side stores), every bench request
    shape once as a single request (many spans) and once inside a batch, against the CPU oracle on the same arrays; plus the
    index cut into two doc-range shards."""
    import veloci_amd
    from veloci_amd import synth
    from oracle import binding as O
    from parity import assert_same
    spec = synth.SynthSpec(num_docs=100_000_000, num_terms=20_000, triples=2, background_terms=0)
    data, meta = synth.generate(spec)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    a, b = list(meta.triples[0]), list(meta.triples[1])
    reqs = [synth.req_and(a), synth.req_or(a), synth.req_single(a[0]), synth.req_single(a[2], top=100), synth.req_and_phrase_locality(a),
            synth.req_and_of_ors([a[0], a[1]], [a[2], b[2]]), dict(synth.req_and(a[1:]), facets=[{"field": "cat"}, {"field": "tags[]", "top": 5}]),
            {"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in a + b]}}, "top": 10},
            dict(synth.req_or(a[1:]), filter={"search": {"path": "body", "terms": [b[2]]}}, skip=5)]
    wants = [ora.search_json(json.dumps(r)) for r in reqs]
    exact = lambda r: "boost" not in r
    for r, w in zip(reqs, wants):
        assert_same(r, veloci_amd.search(r, idx), w, exact_scores=exact(r))
    for r, g, w in zip(reqs, veloci_amd.search_batch(reqs * 8, idx), wants * 8):
        assert_same(r, g, w, exact_scores=exact(r))
    for r, g, w in zip(reqs, _search_batch_over_shards(data, reqs, 2), wants):
        assert not isinstance(g, Exception), (str(g), json.dumps(r))
        assert_same(r, g, w, exact_scores=exact(r))


def test_config4_real_shape_matches_the_oracle():
    """BASELINE configs[3] at its own scale: 10 M docs, a 1 M-term dictionary, lev-2 fuzzy single-term requests (100 vocabulary terms with
    1-2 random edits, the bench's generator) with facets on `cat` and `tags[]` — k_dict_scan over a multi-block grid, CSR offsets at
    T = 1e6, unions of many posting lists (two-level when > 64), facet rows of leaves with ~1e6 hits — against the oracle, single and batched."""
    import veloci_amd
    from veloci_amd import synth
    from oracle import binding as O
    from parity import assert_same
    import bench
    spec = synth.SynthSpec(num_docs=10_000_000, num_terms=1_000_000, triples=4, with_t2t=False, with_phrase=False, with_boost=False, with_facets=True,
                           background_terms=2000)
    data, meta = synth.generate(spec)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    pool = [t for tr in meta.triples for t in tr] + list(meta.background)
    qterms = bench.edited_terms(pool, 100)
    reqs = [{"search_req": {"search": {"path": "body", "terms": [t], "levenshtein_distance": 2}}, "top": 10, "facets": [{"field": "cat"}, {"field": "tags[]"}]}
            for t in qterms]
    wants = [ora.search_json(json.dumps(r)) for r in reqs]
    assert sum(w.num_hits > 0 for w in wants) >= 50 and max(w.num_hits for w in wants) >= 200_000  # the shape is exercised
    for r, w in list(zip(reqs, wants))[:25]:
        assert_same(r, veloci_amd.search(r, idx), w)
    for r, g, w in zip(reqs, veloci_amd.search_batch(reqs, idx), wants):
        assert_same(r, g, w)
    # the same probes inside trees: an AND of two fuzzy leaves, and a fuzzy leaf under a filter
    tree = [{"search_req": {"and": {"queries": [{"search": {"path": "body", "terms": [qterms[i]], "levenshtein_distance": 2}},
                                                 {"search": {"path": "body", "terms": [qterms[i + 1]], "levenshtein_distance": 2}}]}}, "top": 10}
            for i in range(0, 20, 2)]
    for r, g in zip(tree, veloci_amd.search_batch(tree, idx)):
        assert_same(r, g, ora.search_json(json.dumps(r)))


def test_degenerate_inputs(corpus):
    """Empty batch, top 0, skip beyond the hits, a one-document index, an index without documents, a shard that holds no posting of
    the query — against the oracle where it has an answer."""
    import veloci_amd
    from veloci_amd import mini_indexer, synth
    from oracle import binding as O
    from parity import assert_same
    data, meta, idx, ora = corpus
    assert veloci_amd.search_batch([], idx) == []
    a = list(meta.triples[0])
    for req in (dict(synth.req_and(a), top=0), dict(synth.req_or(a), top=0, skip=3), dict(synth.req_and(a), skip=10**6),
                dict(synth.req_single(a[0]), top=1, skip=10**7), dict(synth.req_and(a), top=0, facets=[{"field": "cat"}])):
        assert_same(req, veloci_amd.search(req, idx), ora.search_json(json.dumps(req)))
    # facet with top: null reports every counted value (facet.rs:19-23); more than 1024 entries are ranked on the host from the device's histogram
    req = dict(synth.req_and(a[:2]), facets=[{"field": "cat", "top": None}])
    got, want = veloci_amd.search(req, idx), ora.search_json(json.dumps(req))
    assert_same(req, got, want)
    assert len(dict(got.facets)["cat"]) > 10
    for req in (dict(synth.req_or(a), facets=[{"field": "tags[]", "top": None}]), dict(synth.req_or(a), facets=[{"field": "tags[]", "top": 3000}, {"field": "cat", "top": 3}]),
                dict(synth.req_and(a[:2]), facets=[{"field": "cat"}, {"field": "tags[]", "top": 1025}])):
        got, want = veloci_amd.search(req, idx), ora.search_json(json.dumps(req))
        assert_same(req, got, want)
    assert len(dict(got.facets)["tags[]"]) > 0
    got = veloci_amd.search(dict(synth.req_or(a), facets=[{"field": "tags[]", "top": None}]), idx)
    assert len(dict(got.facets)["tags[]"]) > 1024
    # one document / no document
    for docs in ([{"title": "lonely word", "tags": ["x"]}], []):
        d1, info = mini_indexer.build_index(docs, {"title": {"fulltext": {"tokenize": True}}, "tags[]": {"facet": True}})
        i1 = veloci_amd.Index(d1, device=0)
        o1 = O.OracleIndex(d1.num_anchors)
        d1.load_into(o1)
        for req in ({"search_req": {"search": {"path": "title", "terms": ["lonely"]}}, "facets": [{"field": "tags[]"}]},
                    {"search_req": {"and": {"queries": [{"search": {"path": "title", "terms": ["lonely"]}}, {"search": {"path": "title", "terms": ["word"]}}]}}},
                    {"search_req": {"search": {"path": "title", "terms": ["lonly"], "levenshtein_distance": 1}}},
                    {"search_req": {"search": {"path": "title", "terms": ["absent"]}}}):
            js = json.dumps(req)
            try:
                want = o1.search_json(js)
            except O.OracleError as e:
                with pytest.raises(veloci_amd.VelociError) as g:
                    veloci_amd.search(req, i1)
                assert str(g.value) == str(e), js
                continue
            assert_same(req, veloci_amd.search(req, i1), want)
    # a shard whose doc range holds none of the sparse probe's postings still merges to the unsharded answer
    probe = meta.extra_probes[0]  # df 1000 of 300k docs
    reqs = [synth.req_single(probe), dict(synth.req_and([a[0], probe])), dict(synth.req_or([probe, a[2]]), top=20)]
    for r, g in zip(reqs, _search_batch_over_shards(data, reqs, 7)):
        assert not isinstance(g, Exception), (str(g), json.dumps(r))
        assert_same(r, g, ora.search_json(json.dumps(r)))


def test_deep_paging_beyond_one_scan(big_corpus):
    """top + skip above what one scan ranks (1024): the request runs as a chain of pages, each scan ranking only what lies below
    the previous page's last key; window, num_hits and facets equal the oracle's.  Through vq_search, vq_search_batch and the
    flat entry point, and over doc-range shards, where the merge's caller pages."""
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    data, meta, idx, ora = big_corpus
    a, b = list(meta.triples[0]), list(meta.triples[1])
    shapes = [synth.req_and(a), synth.req_or(a), synth.req_single(a[1]), synth.req_and_phrase_locality(a),
              {"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in a + b[:2]]}}},
              dict(synth.req_and(a[:2]), facets=[{"field": "cat", "top": 4}])]
    reqs = []
    for i, shape in enumerate(shapes):
        for top, skip in ((1500, 0), (10, 3000), (2000, 1000), (7, 1024), (1025, 0), (5, 10**7)):
            if (i + top + skip) % 2 == 0 or skip > 10**6:
                reqs.append(dict(shape, top=top, skip=skip))
    # fewer hits than top + skip asks for: the LAST page is partial (5000 hits: four full pages and 904 more)
    reqs += [synth.req_single(meta.extra_probes[0], top=6000), dict(synth.req_single(meta.extra_probes[0], top=2000), skip=4500),
             dict(synth.req_single(meta.extra_probes[0], top=10, boost=[{"path": "pop", "boost_fun": "Multiply", "param": 1.0}]), skip=4995)]
    wants = [ora.search_json(json.dumps(r)) for r in reqs]
    for r, w in zip(reqs[::3] + reqs[-3:], wants[::3] + wants[-3:]):
        assert_same(r, veloci_amd.search(r, idx), w)
    for r, g, w in zip(reqs, veloci_amd.search_batch(reqs, idx), wants):
        assert_same(r, g, w)
    plain = [(r, w) for r, w in zip(reqs, wants) if "facets" not in r and r["top"] <= 2000]
    num_hits, counts, ids, scores, status = veloci_amd.search_batch_flat([r for r, _ in plain], idx, stride=2000)
    assert not status.any()
    for k, (r, w) in enumerate(plain):
        assert int(num_hits[k]) == w.num_hits and list(ids[k, :counts[k]]) == list(w.ids), json.dumps(r)
        assert np.array_equal(scores[k, :counts[k]].view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32))
    # the step functions on an index that answers alone page as well (over shards they decline: the merge's caller pages, below)
    from veloci_amd import dist as vdist
    sb = veloci_amd.RequestBatch([veloci_amd.Request(r) for r, _ in plain])
    num_hits, counts, ids, scores, status = vdist.shard_step_end(vdist.shard_step_begin(idx, sb), 2000)
    assert not status.any()
    for k, (r, w) in enumerate(plain):
        assert int(num_hits[k]) == w.num_hits and list(ids[k, :counts[k]]) == list(w.ids), json.dumps(r)
        assert np.array_equal(scores[k, :counts[k]].view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32))
    # more ranked hits than 64 scans reach: declined, not truncated
    with pytest.raises(veloci_amd.VelociError) as e:
        veloci_amd.search(dict(synth.req_or(a), top=10, skip=70000), idx)
    assert e.value.kind == "Unsupported" and "ranked hits" in str(e.value)
    # over doc-range shards the caller of the merge pages (vq_result_is_page / vq_request_page_after): every page is one more exchange round
    sharded = reqs[:12] + reqs[-3:] + [synth.req_and(a), dict(synth.req_or(a), top=10, skip=70000)]
    got = _search_batch_over_shards(data, sharded, 2)
    for r, g in zip(sharded[:-1], got[:-1]):
        assert not isinstance(g, Exception), (str(g), json.dumps(r))
        assert_same(r, g, ora.search_json(json.dumps(r)))
    assert isinstance(got[-1], veloci_amd.VelociError) and got[-1].kind == "Unsupported" and "ranked hits" in str(got[-1])
    from veloci_amd.dist import search_shards_local
    N = data.num_anchors
    parts = [veloci_amd.Index(data, device=0, doc_lo=0, doc_hi=N // 4), veloci_amd.Index(data, device=0, doc_lo=N // 4, doc_hi=N)]
    plain = [r for r in reqs if "phrase_boosts" not in r and "text_locality" not in r][:8]
    for r, g in zip(plain, search_shards_local(parts, plain)):
        assert_same(r, g, ora.search_json(json.dumps(r)))


def test_wide_nodes_up_to_16_operands(corpus):
    """The query generator ORs one leaf per (term, field): and / or nodes take up to 16 operands (DOp::child_slot / and_order)."""
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    data, meta, idx, ora = corpus
    terms = [t for tri in meta.triples for t in tri] + list(meta.extra_probes) + list(meta.background[:12])
    leaf = lambda t, **kw: {"search": dict({"path": "body", "terms": [t]}, **kw)}
    reqs = [
        {"search_req": {"or": {"queries": [leaf(t) for t in terms[:12]]}}, "top": 20},
        {"search_req": {"or": {"queries": [leaf(t) for t in terms[:16]]}}, "top": 10, "skip": 3},
        {"search_req": {"or": {"queries": [leaf(t, boost=1.0 + 0.25 * i) for i, t in enumerate(terms[3:16])]}}},
        {"search_req": {"and": {"queries": [{"or": {"queries": [leaf(t) for t in terms[:9]]}}, {"or": {"queries": [leaf(t) for t in terms[9:16]]}}]}}, "top": 15},
        {"search_req": {"or": {"queries": [{"and": {"queries": [leaf(terms[0]), leaf(terms[1])]}}] + [leaf(t) for t in terms[2:14]]}}},
        {"search_req": {"and": {"queries": [leaf(meta.triples[0][0])] + [{"or": {"queries": [leaf(t), leaf(meta.triples[0][1])]}} for t in terms[6:16]]}}},
        {"search_req": {"or": {"queries": [leaf(t) for t in terms[:14]]}}, "filter": {"or": {"queries": [leaf(t) for t in terms[4:15]]}}, "facets": [{"field": "cat"}]},
    ]
    for req in reqs:
        assert_same(req, veloci_amd.search(req, idx), ora.search_json(json.dumps(req)))
    for req, g in zip(reqs, veloci_amd.search_batch(reqs, idx)):
        assert_same(req, g, ora.search_json(json.dumps(req)))
    with pytest.raises(veloci_amd.VelociError) as e:
        veloci_amd.search({"search_req": {"or": {"queries": [leaf(t) for t in terms[:17]]}}}, idx)
    assert e.value.kind == "Unsupported"  # (declined, never truncated: the evaluation stack holds 16 operands)


def test_query_generator_shapes_on_the_wide_kernel(corpus, big_corpus):
    """k_scan_wide: 5-16 single-list leaves in trees of depth <= 2 — the shapes of the reference's query generator (one leaf per term and
    field, src/query_generator.rs:175-246): flat ORs / ANDs, a root over AND / OR groups, dense (bitmap) and sparse (scattered) lists mixed,
    leaf boosts, repeated terms sharing an OR slot, top / skip, single requests (many spans) and batches; on a 300 k- and a 4 M-doc corpus."""
    import veloci_amd
    from parity import assert_same
    for data, meta, idx, ora in (corpus, big_corpus):
        tri = [list(t) for t in meta.triples]
        terms = [t for t3 in tri for t in t3] + list(meta.extra_probes) + list(meta.background[:10])
        leaf = lambda t, **kw: {"search": dict({"path": "body", "terms": [t]}, **kw)}
        OR = lambda qs: {"or": {"queries": qs}}
        AND = lambda qs: {"and": {"queries": qs}}
        reqs = [
            {"search_req": OR([leaf(t) for t in terms[:8]]), "top": 10},
            {"search_req": AND([OR([leaf(t) for t in terms[:4]]), OR([leaf(t) for t in terms[4:8]])]), "top": 10},
            {"search_req": OR([leaf(t) for t in terms[:5]]), "top": 3, "skip": 2},
            {"search_req": OR([AND([leaf(tri[0][0]), leaf(tri[0][1]), leaf(tri[0][2])]), AND([leaf(tri[1][0]), leaf(tri[1][1]), leaf(tri[1][2])])]), "top": 25},
            {"search_req": AND([leaf(tri[0][0]), OR([leaf(t) for t in terms[3:7]]), OR([leaf(t) for t in terms[7:10]])]), "top": 10},
            {"search_req": OR([leaf(t, boost=0.5 + 0.3 * i) for i, t in enumerate(terms[2:15])]), "top": 40},
            {"search_req": OR([leaf(tri[0][0]), leaf(tri[0][0].upper()), leaf(tri[0][1]), leaf(tri[0][2]), leaf(tri[1][0]), leaf(tri[1][1])]), "top": 10},  # one term twice: a shared slot
            {"search_req": AND([leaf(t) for t in (tri[0] + tri[1][:2])]), "top": 10},
            {"search_req": OR([OR([leaf(tri[0][0]), leaf(tri[0][1])]), AND([leaf(tri[0][2]), leaf(tri[1][0])]), leaf(tri[1][1]), leaf(terms[-1]), leaf(terms[-2])]), "top": 12},
            {"search_req": OR([leaf(t) for t in terms[6:6 + 16]][:16]), "top": 10},
            {"search_req": AND([OR([leaf(t) for t in terms[:8]]), OR([leaf(t) for t in terms[8:16]])]), "top": 1000},
        ]
        wants = [ora.search_json(json.dumps(r)) for r in reqs]
        for r, w in zip(reqs, wants):
            assert_same(r, veloci_amd.search(r, idx), w)
        for r, g, w in zip(reqs * 3, veloci_amd.search_batch(reqs * 3, idx), wants * 3):
            assert_same(r, g, w)
        got = _search_batch_over_shards(data, reqs, 2)
        for r, g, w in zip(reqs, got, wants):
            assert not isinstance(g, Exception), (str(g), json.dumps(r))
            assert_same(r, g, w)


def test_token_value_boost_shapes_the_term_scores():
    """RequestSearchPart.token_value (search_field.rs:391-395): a boost column keyed by TERM id (token_values store, built as
    create/token_values_to_tokens.rs does for the reference's test corpus: "Begeisterung" -> 20 on meanings.ger[]) multiplies /
    adds into the matched terms' scores before their postings are read."""
    import veloci_amd
    from veloci_amd import mini_indexer
    from oracle import binding as O
    from parity import assert_same
    c = refcases.load()["corpora"]["test_all"]
    docs = refcases.corpus_docs(c)
    tv = ([{"text": "Begeisterung", "value": 20}, {"text": "welle", "value": 3.5}, {"text": "nothere", "value": 9}, {"text": "der", "value": None}], "meanings.ger[]")
    data, info = mini_indexer.build_index(docs, c["indices"], token_values=tv)
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    ran = 0
    for term, lev in (("begeisterung", 0), ("begeisterung", 1), ("welle", 0), ("majestät", 1), ("der", 0)):
        for tvb in ({"path": "meanings.ger[]", "boost_fun": "Multiply", "param": 0}, {"path": "meanings.ger[]", "boost_fun": "Log10", "param": 1},
                    {"path": "meanings.ger[]", "boost_fun": "Add", "param": 2, "skip_when_score": [10.0]}, {"path": "meanings.ger[]", "expression": "$SCORE * 2"}):
            leaf = {"path": "meanings.ger[]", "terms": [term], "levenshtein_distance": lev, "token_value": tvb}
            for req in ({"search_req": {"search": leaf}},
                        {"search_req": {"or": {"queries": [{"search": leaf}, {"search": {"path": "meanings.eng[]", "terms": ["will"]}}]}}, "top": 20}):
                want = ora.search_json(json.dumps(req))
                assert_same(req, veloci_amd.search(req, idx), want, exact_scores=True)
                ran += 1
    assert ran == 40
    with pytest.raises(veloci_amd.VelociError) as e:  # a field without token values: the reference fails on the missing index
        veloci_amd.search({"search_req": {"search": {"path": "meanings.eng[]", "terms": ["will"], "token_value": {"path": "meanings.eng[]", "boost_fun": "Multiply"}}}}, idx)
    with pytest.raises(O.OracleError) as eo:
        ora.search_json(json.dumps({"search_req": {"search": {"path": "meanings.eng[]", "terms": ["will"], "token_value": {"path": "meanings.eng[]", "boost_fun": "Multiply"}}}}))
    assert str(e.value) == str(eo.value)


def test_random_requests_on_synthetic_corpus_match_the_oracle(corpus):
    """Random trees over the 300k-doc synthetic corpus: dense lists (bitmap images), several spans per query, OR pruning, count pre-passes."""
    _random_synthetic(corpus, n_requests=240, seed=int(os.environ.get("VQ_TEST_SEED", "991")))


def test_random_requests_over_three_shards_match_the_oracle(corpus):
    """The same generator, the index cut into three doc-range shards on one GPU, partials gathered and merged.
    Result sizes and merged list lengths are summed over the shards through the vq_index_set_allreduce hook."""
    _random_synthetic(corpus, n_requests=160, seed=int(os.environ.get("VQ_TEST_SEED", "3003")), shards=3)


def _search_batch_over_shards(data, reqs, shards):
    """The index cut into `shards` doc ranges on one GPU; the shards run in threads and sum the few numbers some requests need over
    all shards (vq_index_set_allreduce); partial buffers are gathered and merged."""
    import veloci_amd
    import torch
    from veloci_amd.dist import device_view
    N = data.num_anchors
    cuts = [N * i // shards for i in range(shards + 1)]
    parts = [veloci_amd.Index(data, device=0, doc_lo=cuts[i], doc_hi=cuts[i + 1]) for i in range(shards)]
    # the shards run in threads and sum the few numbers some requests need over all shards (vq_index_set_allreduce)
    import threading
    barrier = threading.Barrier(shards)
    slots = [None] * shards
    totals = [None]

    def make_hook(rank):
        def hook(values):
            slots[rank] = values.copy()
            barrier.wait()
            if rank == 0:
                totals[0] = np.sum(np.stack(slots), axis=0, dtype=np.uint64)
            barrier.wait()
            values[:] = totals[0]
            barrier.wait()
        return hook

    from veloci_amd.dist import exchange_local
    from veloci_amd.search import complete_deep_pages, _as_request

    def one_round(round_reqs):
        pbs = [None] * shards
        errs = []

        def run(rank):
            try:
                parts[rank].set_allreduce(make_hook(rank))
                pbs[rank] = veloci_amd.PartialBatch(parts[rank], round_reqs)
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(shards)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        assert len({pb.nbytes for pb in pbs}) == 1
        g = exchange_local(pbs)
        res = pbs[0].merge(g.data_ptr(), shards, raise_on_error=False)
        for pb in pbs[1:]:
            pb.merge(None, 1, raise_on_error=False)
        return res

    parsed = [_as_request(r) for r in reqs]
    got = one_round(parsed)
    if any(getattr(g, "is_page", False) for g in got):  # requests reaching beyond one scan's ranking: further rounds over all shards
        got = complete_deep_pages(parsed, got, one_round)
    return got


def _random_synthetic_requests(meta, n_requests, seed, flat=False):
    """flat: only what the flat entry points report (no facets, top + skip within one scan's ranking)"""
    rng = np.random.default_rng(seed)
    pool = [t for tri in meta.triples for t in tri] + list(meta.extra_probes) + list(meta.background[:25])

    def leaf():
        part = {"path": "body", "terms": [pool[int(rng.integers(0, len(pool)))]]}
        r = rng.random()
        if r < 0.08:
            part["terms"] = [part["terms"][0][:3]]
            part["starts_with"] = True
        elif r < 0.16:
            part["levenshtein_distance"] = 1
        if rng.random() < 0.15:
            part["boost"] = float(rng.choice([0.5, 2.0]))
        return {"search": part}

    def tree(d):
        if d >= 2 or rng.random() < 0.35:
            return leaf()
        wide = d >= 1 and rng.random() < 0.15  # the query generator's one-leaf-per-term-and-field nodes: up to 8 operands here, all leaves
        return {("and" if rng.random() < 0.5 else "or"): {"queries": [tree(d + 1) for _ in range(int(rng.integers(5, 9) if wide else rng.integers(2, 5)))]}}

    reqs = []
    for _ in range(n_requests):
        req = {"search_req": tree(0), "top": int(rng.choice([1, 10, 40]))}
        if not flat and rng.random() < 0.06:  # deep paging: beyond what one scan ranks
            req["top"], req["skip"] = int(rng.choice([10, 1200])), int(rng.choice([0, 1100, 2500]))
        if flat and rng.random() < 0.2:
            req["skip"] = int(rng.choice([1, 7]))
        if rng.random() < 0.25:
            req["filter"] = tree(1)
        if rng.random() < 0.3:
            req["boost"] = [{"path": "pop", "boost_fun": str(rng.choice(["Multiply", "Add", "Replace"])), "param": 1.0}]
        if rng.random() < 0.3:
            a, b = (meta.triples[int(rng.integers(0, len(meta.triples)))][i] for i in (0, 1))
            req["phrase_boosts"] = [{"search1": {"path": "body", "terms": [a]}, "search2": {"path": "body", "terms": [b]}}]
        if rng.random() < 0.3:
            req["text_locality"] = True
        if not flat and rng.random() < 0.25:
            req["facets"] = [{"field": str(rng.choice(["cat", "tags[]"])), "top": int(rng.choice([3, 10]))}]
        if rng.random() < 0.2:
            req["boost_term"] = [{"path": "body", "terms": [meta.background[int(rng.integers(0, 30))]], "boost": 3.0}]
        reqs.append(req)
    return reqs


def _random_synthetic(corpus, n_requests, seed, shards=1):
    import veloci_amd
    from parity import assert_same
    data, meta, idx, ora = corpus
    reqs = _random_synthetic_requests(meta, n_requests, seed)
    if shards == 1:
        got = veloci_amd.search_batch(reqs, idx, raise_on_error=False)
    else:
        got = _search_batch_over_shards(data, reqs, shards)
    declined = 0
    for req, g in zip(reqs, got):
        if shards > 1 and isinstance(g, veloci_amd.VelociError) and g.kind == "Unsupported":
            declined += 1
            continue
        assert not isinstance(g, Exception), (str(g), json.dumps(req))
        assert_same(req, g, ora.search_json(json.dumps(req)))
    assert declined == 0, declined


def test_flat_batch_of_1100_mixed_requests_matches_the_oracle(corpus):
    """vq_search_batch_flat with more than 1024 requests — the bench's entry point: two pipelined chunks over two workspaces — against the
    ORACLE row by row (hit counts, ids, score bits): random trees, leaf boosts, prefix / fuzzy leaves, filters, column boosts, phrase pairs,
    text locality, boost_term, skip, plus the bench's own request shapes."""
    import veloci_amd
    from veloci_amd import synth
    data, meta, idx, ora = corpus
    a, b = list(meta.triples[0]), list(meta.triples[1])
    reqs = _random_synthetic_requests(meta, 1000, int(os.environ.get("VQ_TEST_SEED", "1100")), flat=True)
    reqs += [synth.req_and(a), synth.req_or(a), synth.req_single(a[0]), synth.req_and(b[::-1], top=40), synth.req_and([a[0], a[2]], top=1)] * 20
    stride = 48
    num_hits, counts, ids, scores, status = veloci_amd.search_batch_flat(reqs, idx, stride=stride)
    assert len(reqs) >= 1100 and not status.any(), status[status != 0][:5]
    cache = {}
    for i, r in enumerate(reqs):
        key = json.dumps(r, sort_keys=True)
        if key not in cache:
            cache[key] = ora.search_json(json.dumps(r))
        w = cache[key]
        c = int(counts[i])
        assert int(num_hits[i]) == w.num_hits and c == len(w.ids), (i, key, int(num_hits[i]), w.num_hits, c, len(w.ids))
        assert ids[i, :c].tolist() == list(w.ids), (i, key)
        exact = "boost" not in r or all(bq["boost_fun"] not in ("Log10", "Log2") for bq in r["boost"])
        if exact:
            assert np.array_equal(scores[i, :c].view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32)), (i, key, scores[i, :c].tolist(), list(w.scores))


def _check_flat_rows(reqs, out, ora, cache):
    num_hits, counts, ids, scores, status = out
    assert not status.any(), status[status != 0][:5]
    for i, r in enumerate(reqs):
        key = json.dumps(r, sort_keys=True)
        if key not in cache:
            cache[key] = ora.search_json(json.dumps({k: v for k, v in r.items() if k != "facets"}))
        w = cache[key]
        c = int(counts[i])
        assert int(num_hits[i]) == w.num_hits and c == len(w.ids) and ids[i, :c].tolist() == list(w.ids), (i, key, int(num_hits[i]), w.num_hits)
        if "boost" not in r:
            assert np.array_equal(scores[i, :c].view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32)), (i, key)


def test_sharded_step_inside_the_library_two_shards_on_one_gpu(corpus):
    """vq_shard_step_begin / _end (compile -> scans -> exchange -> merge inside the library) over two uneven doc-range shards that live in this
    process, one thread each, with the exchange handed in through vq_comm_init_custom: every row equals the ORACLE's on the unsharded
    corpus — batches of 1100 (two chunks per step on a large shard are forced by VQ_SHARD_CHUNKS in the RCCL test; here one), requests
    with facets (their histograms are summed by the all-reduce hook), and two steps in flight (begin, begin, end, end)."""
    import threading
    import veloci_amd
    from veloci_amd import synth
    from veloci_amd.dist import LocalExchange, shard_step_begin, shard_step_end
    data, meta, idx, ora = corpus
    N = data.num_anchors
    cuts = [0, N * 2 // 5, N]
    parts = [veloci_amd.Index(data, device=0, doc_lo=cuts[i], doc_hi=cuts[i + 1]) for i in range(2)]
    ex = LocalExchange(parts)
    barrier = threading.Barrier(2)
    slots, totals = [None, None], [None]

    def make_hook(rank):
        def hook(values):
            slots[rank] = values.copy()
            barrier.wait()
            if rank == 0:
                totals[0] = np.sum(np.stack(slots), axis=0, dtype=np.uint64)
            barrier.wait()
            values[:] = totals[0]
            barrier.wait()
        return hook

    a = list(meta.triples[0])
    reqs1 = _random_synthetic_requests(meta, 500, 77, flat=True) + [synth.req_and(a), synth.req_or(a), synth.req_single(a[0])] * 30
    reqs1 += [dict(synth.req_and(a[1:]), facets=[{"field": "cat"}, {"field": "tags[]", "top": 5}]), dict(synth.req_single(a[1]), facets=[{"field": "cat", "top": 3}])] * 5
    reqs2 = _random_synthetic_requests(meta, 300, 78, flat=True)
    b1, b2 = veloci_amd.RequestBatch(reqs1), veloci_amd.RequestBatch(reqs2)
    outs, errs = [None, None], []

    def run(rank):
        try:
            parts[rank].set_allreduce(make_hook(rank))
            s1 = shard_step_begin(parts[rank], b1)
            s2 = shard_step_begin(parts[rank], b2)  # two steps in flight
            o1 = shard_step_end(s1, 48)
            o2 = shard_step_end(s2, 48)
            s3 = shard_step_begin(parts[rank], b2)
            o3 = shard_step_end(s3, 48)
            outs[rank] = (o1, o2, o3)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            barrier.abort()
            ex.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    cache = {}
    for rank in range(2):  # every rank holds the merged result
        _check_flat_rows(reqs1, outs[rank][0], ora, cache)
        _check_flat_rows(reqs2, outs[rank][1], ora, cache)
        _check_flat_rows(reqs2, outs[rank][2], ora, cache)


def test_sharded_step_inside_the_library_over_rccl_with_one_rank(corpus):
    """The same step with the library's own RCCL communicator (vq_comm_unique_id / vq_comm_init; one rank: what a 1-GPU box can rehearse):
    ncclAllGather of the partials, ncclAllReduce of the facet histograms and of the sums some requests need before compilation, one and two
    chunks per step, two steps in flight — against the oracle."""
    import ctypes as C
    import veloci_amd
    from veloci_amd import _lib, synth
    from veloci_amd.dist import shard_step_begin, shard_step_end
    data, meta, idx0, ora = corpus
    idx = veloci_amd.Index(data, device=0)
    L = _lib.lib()
    ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(L.vq_comm_unique_id(ident))
    _lib.check(L.vq_comm_init(idx.h, 1, 0, bytes(ident)))
    a = list(meta.triples[0])
    reqs = _random_synthetic_requests(meta, 700, 79, flat=True) + [synth.req_and(a), synth.req_or(a)] * 20
    reqs += [dict(synth.req_and(a[1:]), facets=[{"field": "cat"}, {"field": "tags[]", "top": 5}])] * 4
    batch = veloci_amd.RequestBatch(reqs)
    cache = {}
    s1 = shard_step_begin(idx, batch)
    s2 = shard_step_begin(idx, batch)
    _check_flat_rows(reqs, shard_step_end(s1, 48), ora, cache)
    _check_flat_rows(reqs, shard_step_end(s2, 48), ora, cache)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    out = (np.zeros(len(reqs), np.uint64), np.zeros(len(reqs), np.uint32), np.zeros((len(reqs), 48), np.uint32), np.zeros((len(reqs), 48), np.float32), np.zeros(len(reqs), np.int32))
    _lib.check(L.vq_shard_step_flat(idx.h, batch.arr, batch.n, 48, *[p(x) for x in out]))
    _check_flat_rows(reqs, out, ora, cache)
    _lib.check(L.vq_comm_destroy(idx.h))


def test_sharded_step_error_paths_leave_the_index_usable(corpus):
    """What a caller can get wrong with the step functions: a third step begun while two are in flight, a stride smaller than a request's top,
    a step given up without its end (vq_shard_step_free), a step of zero requests, requests the path declines inside a step — each answered with
    its error (or status), none of them leaving a workspace locked or a merge pending: the next steps still equal the oracle."""
    import ctypes as C
    import veloci_amd
    from veloci_amd import _lib, synth
    from veloci_amd.dist import shard_step_begin, shard_step_end
    data, meta, idx0, ora = corpus
    idx = veloci_amd.Index(data, device=0)
    L = _lib.lib()
    a = list(meta.triples[0])
    reqs = [synth.req_and(a), synth.req_or(a, top=30), synth.req_single(meta.extra_probes[0])] * 40
    batch = veloci_amd.RequestBatch(reqs)
    cache = {}
    s1 = shard_step_begin(idx, batch)
    s2 = shard_step_begin(idx, batch)
    with pytest.raises(veloci_amd.VelociError) as e:
        shard_step_begin(idx, batch)
    assert e.value.kind == "InvalidArgument" and "two steps are in flight" in str(e.value)
    with pytest.raises(veloci_amd.VelociError) as e:  # top 30 does not fit a stride of 10: the step is consumed by the failed end
        shard_step_end(s1, 10)
    assert "stride" in str(e.value)
    _check_flat_rows(reqs, shard_step_end(s2, 32), ora, cache)
    s3 = shard_step_begin(idx, batch)  # given up without an end: its scans are waited for, its workspace is handed on
    L.vq_shard_step_free(s3[0])
    empty = veloci_amd.RequestBatch([])
    out = shard_step_end(shard_step_begin(idx, empty), 10)
    assert len(out[0]) == 0
    # a request the path declines (explain on a part of its tree) and one that fails to compile (unknown field) ride along: statuses, not a failed step
    part = synth.req_single(meta.extra_probes[0])["search_req"]["search"]
    declined = {"search_req": {"or": {"queries": [{"search": dict(part, options={"explain": True})}, {"search": dict(part)}]}}}
    mixed = reqs[:6] + [declined, {"search_req": {"search": {"terms": ["x"], "path": "nosuchfield"}}}]
    mb = veloci_amd.RequestBatch(mixed)
    res = shard_step_end(shard_step_begin(idx, mb), 32)
    assert list(res[4][:6]) == [0] * 6 and res[4][6] != 0 and res[4][7] != 0
    _check_flat_rows(mixed[:6], tuple(x[:6] for x in res), ora, cache)
    for _ in range(2):  # and the pipeline still runs
        s4 = shard_step_begin(idx, batch)
        s5 = shard_step_begin(idx, batch)
        _check_flat_rows(reqs, shard_step_end(s4, 32), ora, cache)
        _check_flat_rows(reqs, shard_step_end(s5, 32), ora, cache)


def test_rccl_collective_path_with_one_rank(corpus, monkeypatch):
    """dist.ShardedSearcher on the `nccl` (= RCCL) backend with a single rank.  First the module's own exchange (VQ_PY_COLLECTIVE=1): the
    scans run on a torch side stream handed to the index (vq_index_set_stream), the packed partial is all-gathered by RCCL as a zero-copy
    uint8 view, the merge reads the gathered buffer, the all-reduce hook goes through RCCL.  Then the default: the library's own
    communicator (the id travels through the process group), steps through vq_shard_step_begin / _end — what the multi-GPU bench does,
    minus the other ranks."""
    import socket
    import torch
    import torch.distributed as dist
    import veloci_amd
    from veloci_amd import synth
    from veloci_amd.dist import ShardedSearcher
    data, meta, idx, ora = corpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        idx2 = veloci_amd.Index(data, device=0)
        monkeypatch.setenv("VQ_PY_COLLECTIVE", "1")
        searcher = ShardedSearcher(idx2, always_collective=True)
        assert searcher.stream is not None and not searcher.native
        t = [list(x) for x in meta.triples]
        reqs = []
        for i in range(96):
            a = t[i % len(t)]
            reqs.append([synth.req_and(a), synth.req_or(a, top=25), synth.req_single(a[i % 3], top=3),
                         dict(synth.req_and(a[:2]), facets=[{"field": "cat", "top": 5}])][i % 4])
        # paged over further collectives; > 1024 facet entries ranked on the host from the all-reduced histogram
        deep = [dict(synth.req_or(t[0]), top=1500, skip=700), dict(synth.req_single(meta.extra_probes[1]), top=10, skip=2500, facets=[{"field": "tags[]", "top": None}])]
        for g, w in zip(searcher.search_batch(reqs[:5] + deep)[5:], veloci_amd.search_batch(deep, idx)):
            assert g.num_hits == w.num_hits and list(g.ids) == list(w.ids) and len(g.ids) > 0
            assert np.array_equal(np.asarray(g.scores, np.float32).view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32))
            assert sorted(map(repr, (g.facets or {}).items())) == sorted(map(repr, (w.facets or {}).items()))
        want = veloci_amd.search_batch(reqs, idx)
        for _ in range(3):  # several batches back to back on the side stream
            got = searcher.search_batch(reqs)
            for g, w in zip(got, want):
                assert g.num_hits == w.num_hits and list(g.ids) == list(w.ids)
                assert np.array_equal(np.asarray(g.scores, np.float32).view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32))
                assert sorted(map(repr, (g.facets or {}).items())) == sorted(map(repr, (w.facets or {}).items()))
        plain = [r for r in reqs if "facets" not in r]
        f_want = veloci_amd.search_batch_flat(plain * 8, idx, stride=25)
        for per_chunk in (False, True, False):  # >= 512 requests: a pipeline of 4 chunks — their partials in the arena and ONE all-gather; or one per chunk
            if per_chunk:
                os.environ["VQ_PER_CHUNK_COLLECTIVE"] = "1"
            else:
                os.environ.pop("VQ_PER_CHUNK_COLLECTIVE", None)
            f_got = searcher.search_batch_flat(plain * 8, stride=25)
            for a, b in zip(f_got, f_want):
                assert np.array_equal(a, b)
        for chunks in (2, 3):
            for a, b in zip(searcher.search_batch_flat(plain * 8, stride=25, chunks=chunks), f_want):
                assert np.array_equal(a, b)
        # chunks with facet histograms fall back to the per-chunk exchange (all-gather + all-reduce each)
        with_facets = (reqs * 6)[:560]
        m_got = searcher.search_batch_flat(with_facets, stride=25)
        m_want = veloci_amd.search_batch_flat(with_facets, idx, stride=25)
        for a, b in zip(m_got, m_want):
            assert np.array_equal(a, b)
        v = np.array([1, 2, 2**40 + 7], dtype=np.uint64)
        searcher._sum_over_ranks(v)
        assert v.tolist() == [1, 2, 2**40 + 7]
        # ---- the default: the exchange inside the library
        monkeypatch.delenv("VQ_PY_COLLECTIVE")
        idx3 = veloci_amd.Index(data, device=0)
        native = ShardedSearcher(idx3, always_collective=True)
        assert native.native
        for a, b in zip(native.search_batch_flat(plain * 8, stride=25), f_want):
            assert np.array_equal(a, b)
        s1, s2 = native.step_begin(with_facets), native.step_begin(plain * 8)  # two steps in flight
        for a, b in zip(native.step_end(s1, 25), m_want):
            assert np.array_equal(a, b)
        for a, b in zip(native.step_end(s2, 25), f_want):
            assert np.array_equal(a, b)
        for g, w in zip(native.search_batch(reqs), want):  # result objects still go through the module's exchange
            assert g.num_hits == w.num_hits and list(g.ids) == list(w.ids)
            assert sorted(map(repr, (g.facets or {}).items())) == sorted(map(repr, (w.facets or {}).items()))
    finally:
        dist.destroy_process_group()


def test_concurrent_searches_from_host_threads(corpus):
    """The reference serves `search` from many rocket worker threads on one Persistence (server/rocket_server.rs:139-145);
    the index handle must take concurrent vq_search / vq_search_batch calls (ctypes drops the GIL during them)."""
    import threading
    import veloci_amd
    from veloci_amd import synth
    from parity import assert_same
    data, meta, idx, ora = corpus
    a, b, c = meta.triples[0]
    d, e, f = meta.triples[1]
    reqs = [synth.req_and([a, b, c]), synth.req_or([a, e]), synth.req_single(meta.extra_probes[1], top=20), synth.req_and_phrase_locality([a, b, c]),
            synth.req_and_of_ors([a, b], [c, d]), {"search_req": {"search": {"path": "body", "terms": [a[:3]], "starts_with": True}}, "top": 5},
            dict(synth.req_or([d, e, f], top=15), facets=[{"field": "cat"}]),
            {"search_req": {"and": {"queries": [{"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in (a, d)]}},
                                                {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in (b, e)]}},
                                                {"search": {"path": "body", "terms": [c]}}]}}}]
    want = [ora.search_json(json.dumps(r)) for r in reqs]
    errors = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for it in range(40):
                if it % 3 == 0:
                    order = [int(x) for x in rng.permutation(len(reqs))]
                    got = veloci_amd.search_batch([reqs[i] for i in order], idx)
                    for i, g in zip(order, got):
                        assert_same(reqs[i], g, want[i], exact_scores=(i != 4))
                else:
                    i = int(rng.integers(0, len(reqs)))
                    assert_same(reqs[i], veloci_amd.search(reqs[i], idx), want[i], exact_scores=(i != 4))
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex)[:2000])

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]


def test_query_generator_requests_match_the_reference_and_the_oracle():
    """the requests `query_generator::search_query` builds for the reference's own tests (tests/golden/reference_query_generator.json:
    every term expanded into an OR over all 14 searched fields) through the C ABI — the reference's assertions, and product == oracle bit for bit."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    import refcases
    import test_reference_integration as T
    fx = T._load_query_generator()
    built = {}
    ran = 0
    for case in fx["cases"]:
        if "request" not in case:
            continue
        if case["corpus"] not in built:
            c = fx["corpora"][case["corpus"]]
            data, docs, info = T.build_fixture_corpus(fx, case["corpus"], c.get("token_values"))
            ora = O.OracleIndex(data.num_anchors)
            data.load_into(ora)
            built[case["corpus"]] = (veloci_amd.Index(data, device=0), ora, docs, info)
        idx, ora, docs, info = built[case["corpus"]]
        request = dict(case["request"])
        res = refcases.check_expectations(case, docs, info, lambda req: veloci_amd.search(req, idx))
        want = ora.search_json(json.dumps(request))
        assert_same(request, res, want)
        assert res.explain_json == (want.explain_json if request.get("explain") else "null"), case["name"]
        request.pop("explain", None)
        for extra in ({"why_found": True}, {"top": 2, "skip": 1}):
            wide = dict(request, **extra)
            assert_same(wide, veloci_amd.search(wide, idx), ora.search_json(json.dumps(wide)))
        ran += 1
    assert ran == 29


def test_explain_records_match_the_reference_and_the_oracle():
    """the Explain records (src/search/result/explain.rs:2-21) of the returned hits — the reference's own assertions
    (tests/golden/reference_explain.json), then the product's JSON == the oracle's, character for character (floats as %.9g of the f32), over
    requests that reach every record-producing path; and what is declined is declined loudly."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    import refcases
    import test_reference_integration as T
    data, docs, info = refcases.build("test_all")
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    idx = veloci_amd.Index(data, device=0)
    for case in T._load_explain()["cases"]:
        res = refcases.check_expectations(case, docs, info, lambda req: veloci_amd.search(req, idx))
        want = ora.search_json(json.dumps(case["request"]))
        assert_same(case["request"], res, want)
        assert res.explain_json == want.explain_json, case["name"]
    reqs = T.explain_requests()
    for req in reqs:
        got, want = veloci_amd.search(req, idx), ora.search_json(json.dumps(req))
        assert_same(req, got, want)
        assert got.explain_json == want.explain_json, json.dumps(req)
    batch = veloci_amd.search_batch(reqs + [{k: v for k, v in reqs[2].items() if k != "explain"}], idx)
    for req, got in zip(reqs, batch):
        assert got.explain_json == ora.search_json(json.dumps(req)).explain_json
    assert batch[-1].explain_json == "null"
    leaf = lambda **kw: {"search": kw}
    declined = [
        {"search_req": {"or": {"queries": [leaf(terms=["will"], path="meanings.eng[]", options={"explain": True}), leaf(terms=["urge"], path="meanings.eng[]")]}}},
        {"search_req": {"or": {"queries": [leaf(terms=["will"], path="meanings.eng[]"), leaf(terms=["urge"], path="meanings.eng[]")]}}, "explain": True,
         "phrase_boosts": [{"search1": {"terms": ["will"], "path": "meanings.eng[]"}, "search2": {"terms": ["urge"], "path": "meanings.eng[]"}}]},
    ]
    for req in declined:
        with pytest.raises(veloci_amd.VelociError) as e:
            veloci_amd.search(req, idx)
        assert e.value.kind == "Unsupported" and "explain" in str(e.value), req


def test_repeated_leaves_inside_one_or_are_fused_like_the_reference_merges_them():
    """The same leaf twice under one OR (and the same term over several fields): the product fuses them into one leaf, the reference runs them as
    separate operands of one term slot — same hits, scores, explain records, why_found terms, facets."""
    import veloci_amd
    from oracle import binding as O
    from parity import assert_same
    data, docs, info = refcases.build("test_all")
    idx = veloci_amd.Index(data, device=0)
    ora = O.OracleIndex(data.num_anchors)
    data.load_into(ora)
    leaf = lambda **kw: {"search": kw}
    eng, ger = "meanings.eng[]", "meanings.ger[]"
    reqs = []
    for extra in ({}, {"explain": True}, {"why_found": True}, {"text_locality": True}, {"facets": [{"field": "tags[]"}]}):
        reqs += [
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["will"], path=eng), leaf(terms=["urge"], path=eng)]}}}, **extra),
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["will"], path=eng)]}}}, **extra),
            dict({"search_req": {"or": {"queries": [leaf(terms=["will"], path=eng, levenshtein_distance=1), leaf(terms=["will"], path=eng, levenshtein_distance=1),
                                                    leaf(terms=["will"], path=ger, levenshtein_distance=1)]}}}, **extra),
            dict({"search_req": {"and": {"queries": [{"or": {"queries": [leaf(terms=["will"], path=eng), leaf(terms=["will"], path=ger), leaf(terms=["will"], path=eng)]}},
                                                     leaf(terms=["will"], path=eng)]}}}, **extra),
        ]
    for r in reqs:
        js = json.dumps(r)
        got, want = veloci_amd.search(r, idx), ora.search_json(js)
        assert_same(r, got, want)
        assert got.explain_json == (want.explain_json if r.get("explain") else "null"), js
        assert {k: sorted(v) for k, v in got.why_found_terms.items()} == {k: sorted(v) for k, v in want.why_found_terms.items()}, js
    for r, g in zip(reqs, veloci_amd.search_batch(reqs, idx)):
        assert_same(r, g, ora.search_json(json.dumps(r)))
