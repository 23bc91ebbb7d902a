"""N>1 path on CPU: two gloo ranks, doc-range shards, all-gather of packed partials, merge.

No GPU here, so each rank's shard engine is the CPU oracle run on that rank's shard of the synthetic index;
what is under test is the sharding scheme and the collective plumbing of veloci_amd.dist: shard ranges, the
shard == slice-of-the-unsharded-index property of the generator, the all-reduce of list lengths, the
all-gather of equal-sized packed partial buffers and the exactness of "per-shard top-k + merge"."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _order_f32(bits):
    bits = np.uint32(bits)
    return np.uint32(~bits) if bits & np.uint32(0x80000000) else np.uint32(bits | np.uint32(0x80000000))


def _pack(results, top_k, hist_sizes):
    """Same split as the device partial: ([u64 hits[nq]][u64 keys[nq*top_k]], [u32 hist[...]])."""
    nq = len(results)
    hits = np.zeros(nq, np.uint64)
    keys = np.zeros(nq * top_k, np.uint64)
    hist = []
    for q, r in enumerate(results):
        hits[q] = r.num_hits
        for j, (d, s) in enumerate(zip(r.ids.tolist(), r.scores.tolist())):
            bits = np.float32(s).view(np.uint32)
            keys[q * top_k + j] = (np.uint64(_order_f32(bits)) << np.uint64(32)) | np.uint64(d)
        for (field, entries), (fname, C, names) in zip(r.facets, hist_sizes):
            h = np.zeros(C, np.uint32)
            for v, c in entries:
                h[names[v]] = c
            hist.append(h)
    # the all-gathered part (hit counts, keys) and the all-reduced part (facet histograms) — DESIGN.md §6
    buf = np.concatenate([hits.view(np.uint8), keys.view(np.uint8)])
    hbuf = np.concatenate([h.view(np.uint8) for h in hist]) if hist else np.zeros(0, np.uint8)
    return torch.from_numpy(buf.copy()), torch.from_numpy(hbuf.copy())


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from veloci_amd import dist as vdist
    from veloci_amd import synth
    from oracle import binding as O

    N = 60_000
    spec = synth.SynthSpec(num_docs=N, num_terms=500, triples=1, extra_probe_dfs=(400,), background_terms=5, cat_values=16, tag_values=64)
    lo, hi = vdist.shard_range(N, rank, world)
    data, meta = synth.generate(spec, doc_lo=lo, doc_hi=hi, device="cpu")
    path = "body.textindex.to_anchor_id_score"
    local_lens = np.diff(data.token_to_anchor_score[path][0].astype(np.int64))
    vdist.all_reduce_global_lens(data)
    global_lens = data.token_to_anchor_score[path][3].astype(np.int64)

    ora = O.OracleIndex(N)
    data.load_into(ora)
    a, b, c = meta.triples[0]
    top_k = 7
    cats = [("cat", 16, {("cat%05d" % i): i for i in range(16)})]
    reqs = [synth.req_and([a, b, c], top=top_k), synth.req_or([a, b], top=top_k), synth.req_single(meta.extra_probes[0], top=top_k)]
    reqs[1]["facets"] = [{"field": "cat", "top": 16}]
    res = [ora.search_json(json.dumps(r)) for r in reqs]
    local, local_hist = _pack(res, top_k, cats)
    gathered = vdist.gather_partials(local)
    assert gathered.numel() == world * local.numel()
    summed_hist = vdist.reduce_histograms(local_hist.clone())  # one all-reduce: every rank ends up with the global counts

    if rank == 0:
        full, _ = synth.generate(spec, device="cpu")
        assert np.array_equal(np.diff(full.token_to_anchor_score[path][0].astype(np.int64)), global_lens), "all-reduced lengths != unsharded lengths"
        # shard == slice of the unsharded index
        fo, fa, fs, _ = full.token_to_anchor_score[path]
        so, sa, ss, _ = data.token_to_anchor_score[path]
        for t in np.nonzero(local_lens)[0][:50]:
            fl = fa[int(fo[t]):int(fo[t + 1])]
            m = (fl >= lo) & (fl < hi)
            assert np.array_equal(fl[m], sa[int(so[t]):int(so[t + 1])])
            assert np.array_equal(fs[int(fo[t]):int(fo[t + 1])][m], ss[int(so[t]):int(so[t + 1])])
        want_ora = O.OracleIndex(N)
        full.load_into(want_ora)
        want = [want_ora.search_json(json.dumps(r)) for r in reqs]
        nq = len(reqs)
        per = local.numel()
        g = gathered.numpy().reshape(world, per)
        hits = np.zeros(nq, np.uint64)
        keys = [[] for _ in range(nq)]
        hist = np.zeros(16, np.uint64)
        for p in range(world):
            hits += g[p, :nq * 8].view(np.uint64)
            k = g[p, nq * 8:nq * 8 + nq * top_k * 8].view(np.uint64).reshape(nq, top_k)
            for q in range(nq):
                keys[q] += [int(x) for x in k[q] if x]
        hist += summed_hist.numpy().view(np.uint32)
        problems = []
        for q in range(nq):
            merged = sorted(keys[q], reverse=True)[:top_k]
            ids = [x & 0xFFFFFFFF for x in merged]
            if int(hits[q]) != want[q].num_hits:
                problems.append(f"q{q} hits {hits[q]} != {want[q].num_hits}")
            if ids != want[q].ids.tolist():
                problems.append(f"q{q} ids {ids} != {want[q].ids.tolist()}")
        wf = dict(want[1].facets)["cat"]
        got_counts = sorted([int(x) for x in hist if x], reverse=True)
        if got_counts != [c for _, c in wf]:
            problems.append(f"facet counts {got_counts} != {[c for _, c in wf]}")
        with open(out_path, "w") as f:
            json.dump(problems, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_search(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "problems.json")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    with open(out) as f:
        problems = json.load(f)
    assert problems == []


def test_shard_range_partitions_everything():
    from veloci_amd.dist import shard_range
    for n in (1, 7, 100_000_000, 2**32 - 2):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_sharded_deep_paging_loop_pages_until_reach_and_applies_skip_and_top():
    """`search.complete_deep_pages` (the host loop a caller of the sharded merge runs) against a ranked list held in Python: page 0
    arrives flagged, every further round is asked for with the last hit's (score, id), paging stops at top + skip, at the end of the hits, or
    at an empty page; a failing round fails only its request; beyond 65 536 ranked hits the request is declined."""
    import numpy as np
    import importlib
    S = importlib.import_module("veloci_amd.search")  # (the package re-exports a function of the same name)
    from veloci_amd._lib import VelociError

    class FakeRequest(S.Request):
        def __init__(self, top, skip, ranked, fail_at=None):
            self.h = None
            self.top, self.skip, self.ranked, self.fail_at = top, skip, ranked, fail_at
            self.after = None

        def top_skip(self):
            return self.top, self.skip

        def page_after(self, score, doc_id):
            page = FakeRequest(self.top, self.skip, self.ranked, self.fail_at)
            page.after = (float(score), int(doc_id))
            return page

    def page_of(req):
        start = 0 if req.after is None else 1 + next(i for i, (s, d) in enumerate(req.ranked) if (float(s), int(d)) == req.after)
        if req.fail_at is not None and start >= req.fail_at:
            return VelociError(5, "Device: injected")
        rows = req.ranked[start:start + 1024]
        res = S.SearchResult(len(req.ranked), np.array([d for _, d in rows], np.uint32), np.array([s for s, _ in rows], np.float32), None, 0)
        return res

    rounds = []

    def run_page(pages):
        rounds.append(len(pages))
        return [page_of(p) for p in pages]

    def ranked(n):
        return [(np.float32(1000.0 - 0.01 * i), 7 * i + 3) for i in range(n)]

    cases = [(1500, 0, 5000), (10, 3000, 5000), (2000, 1000, 2500), (7, 1024, 5000), (1025, 0, 1025), (5, 10**7, 5000), (6000, 0, 5000), (10, 4995, 5000), (3000, 0, 1024)]
    reqs = [FakeRequest(t, s, ranked(n)) for t, s, n in cases] + [FakeRequest(4000, 0, ranked(5000), fail_at=2048), FakeRequest(10, 70000, ranked(80000)), FakeRequest(10, 0, ranked(50))]
    first = [page_of(r) for r in reqs]
    for r, res in zip(reqs, first):
        res.is_page = r.top + r.skip > 1024
    out = S.complete_deep_pages(reqs, first, run_page)
    for (t, s, n), res in zip(cases, out):
        want = ranked(n)[s:s + t]
        assert list(res.ids) == [d for _, d in want] and not res.is_page, (t, s, n, len(res.ids))
        assert np.array_equal(res.scores, np.array([x for x, _ in want], np.float32))
    assert isinstance(out[len(cases)], VelociError) and out[len(cases)].kind == "Device"
    assert isinstance(out[len(cases) + 1], VelociError) and out[len(cases) + 1].kind == "Unsupported"
    assert len(out[-1].ids) == 50
    assert rounds and max(rounds) <= len(reqs)  # every round carries only the requests that still page


def _run_gloo_step_drivers(scenario):
    """two processes of tests/native/gloo_step_driver.py over the host-stub build -> their stdout / stderr tails"""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "veloci_amd", "csrc"), "-j6", "hoststub"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    so = os.path.join(ROOT, "veloci_amd", "_host_stub", "libveloci_host_stub.so")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, VQ_LIB=so, VQ_STUB_NOOP_LAUNCH="1", VQ_HOST_THREADS="2", RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "native", "gloo_step_driver.py"), scenario], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank of the in-library step hung")
        outs.append((p.returncode, o, e))
    return outs


@pytest.mark.parametrize("scenario", ["ok", "fail"])
def test_in_library_step_with_two_gloo_ranks(scenario):
    """vq_shard_step_begin / _end driven from TWO processes through vq_comm_init_custom (the exchange: gloo), on the host-stub build of the
    library — pipelined steps with and without facet histograms issue the same collectives in the same order on both ranks (`ok`); when one
    rank's step fails between its collectives, BOTH ranks get an error for that step instead of one of them waiting for ever, the communicator
    is down afterwards on both, and the shard answers alone again once it is destroyed (`fail`).  SURVEY.md 8(e); capi.cpp vq_shard_step_*."""
    outs = _run_gloo_step_drivers(scenario)
    for rank, (rc, o, e) in enumerate(outs):
        assert rc == 0 and "GLOO_STEP_DRIVER_OK" in o, f"rank {rank} rc {rc}\n{o[-1500:]}\n{e[-4000:]}"
    stats = [json.loads(o.split("GLOO_STEP_DRIVER_OK ", 1)[1]) for _, o, _ in outs]
    assert stats[0]["steps"] == stats[1]["steps"] >= 5 and stats[1]["collectives"] >= 5
    assert stats[0]["collectives"] == stats[1]["collectives"] + (1 if scenario == "fail" else 0)  # (rank 0 entered the failed step's exchange, rank 1 never got there)
    if scenario == "fail":
        assert "injected failure" in stats[1]["failed_step_error"] and "all-gather failed" in stats[0]["failed_step_error"]
