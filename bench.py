#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): queries/sec of 3-term AND on a 100 M-doc synthetic index.

    python bench.py --gpus N --steps K --warmup W

A step is one batch of --batch (default 1024) 3-term AND queries, top 10, executed through the C ABI
(`vq_search_batch_flat` / the sharded partial + merge path).  The index is fixed at --docs (default 1e8)
documents and sharded by doc-id range over the N ranks (strong scaling); per-shard top-k are merged after
one RCCL all-gather per batch.  `--gpus N` without a launcher starts the N ranks itself (a child
`torch.distributed.run`, spawned before anything touches the GPU).  Rank 0 prints ONE JSON line.

roofline    : per kernel, from HIP events the library records around every launch on its stream inside the timed
              region (vq_profile_json).  `achieved` = bytes THIS data layout has to move per launch (bitmap words of
              dense lists, 4 B per id of scattered lists, 6 B per streamed posting, per-hit gathers as counted by the
              kernels, 8 B per key) / mean launch time — a real fraction of the HBM peak, <= 1.  The SURVEY.md 8(d)
              posting-streaming accounting (6 B per posting of every list) is kept as `algorithmic_equiv`; it exceeds
              the peak where the layout reads dense lists as bitmaps, and says how fast a design that streams postings
              would have to read to keep up.
cpu_baseline: the CPU oracle (C++ restatement of the reference algorithm, `kind: port`) timed on this host on a
              bounded sample of the same workload (rank 0, N=1 only).
configs     : (N=1) the same accounting for BASELINE configs #2 (1 M docs, and the 100 M-doc index), #3 and #4,
              a few steps each, so that every named shape has a driver-run line; --no-extra skips them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0        # measured float4 copy on MI355X (same guide): what a streaming kernel can reach
DEFAULT_BATCH = 1024          # queries per step
DEFAULT_TRIPLES = 512         # distinct probe triples: a step runs as two launches of 512 queries (VQ_SHARD_CHUNKS=2), every query of a launch reads its own
                              # three lists.  (1024 triples would need ~320 GB of host memory to generate and stage: the GPU boxes' limit is 300 GiB)

WORKLOADS = {"and": "3-term AND", "or": "3-term OR", "single": "single-term scan", "config2": "single-term scan over df 1e3 / 1e5 / 1e6 (config #2)",
             "config3": "3-term AND + 2 phrase pairs + text locality",
             "and_of_ors": "AND(OR,OR) + Log10 boost + phrase + locality", "mix": "40% AND / 40% OR / 20% AND(OR,OR)+boost+phrase+locality",
             "config4": "lev-2 fuzzy term + facets (cat, tags[])", "or8": "flat OR over 8 terms", "and_of_or4": "AND of two 4-term ORs"}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start one rank per GPU as a child torch.distributed.run (nothing in this
    process has touched the GPU; it only waits for the child and passes its exit code on)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting", args.gpus, "ranks:", " ".join(cmd[2:8]))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def edited_terms(pool, count, seed=4):
    """vocabulary terms with 1-2 random edits each; distinct"""
    import numpy as np
    rng = np.random.default_rng(seed)
    alphabet = "abcdefghijklmnopqrstuvwxyz"

    def edit(w):
        w = list(w)
        for _ in range(int(rng.integers(1, 3))):
            op = int(rng.integers(0, 3))
            pos = int(rng.integers(0, len(w)))
            if op == 0 and len(w) > 3:
                del w[pos]
            elif op == 1:
                w.insert(pos, alphabet[int(rng.integers(0, 26))])
            else:
                w[pos] = alphabet[int(rng.integers(0, 26))]
        return "".join(w)

    out, seen = [], set()
    while len(out) < count:
        t = edit(pool[int(rng.integers(0, len(pool)))])
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


def make_requests(workload, meta, batch, probes=1024, tri_limit=None):
    from veloci_amd import synth
    n_tri = min(len(meta.triples), tri_limit or len(meta.triples))  # (tri_limit: cycle over the first few triples only — launches that hold every list several times)
    tri = lambda i: list(meta.triples[i % n_tri])
    if workload == "and":
        return [synth.req_and(tri(i), top=10) for i in range(batch)]
    if workload == "or":
        return [synth.req_or(tri(i), top=10) for i in range(batch)]
    if workload == "single":
        return [synth.req_single(meta.triples[i % len(meta.triples)][i // len(meta.triples) % 3], top=10) for i in range(batch)]
    if workload == "config2":
        return [synth.req_single(meta.extra_probes[i % len(meta.extra_probes)], top=10) for i in range(batch)]
    if workload == "config4":
        pool = [t for tr in meta.triples for t in tr] + list(meta.background)
        qterms = edited_terms(pool, min(probes, batch))
        return [{"search_req": {"search": {"path": "body", "terms": [qterms[i % len(qterms)]], "levenshtein_distance": 2}}, "top": 10,
                 "facets": [{"field": "cat"}, {"field": "tags[]"}]} for i in range(batch)]
    if workload in ("or8", "and_of_or4"):
        # shapes of the reference's query generator (one term expanded over several fields, src/query_generator.rs): a flat OR over 8
        # leaves, and an AND of two 4-leaf ORs
        def terms8(i):
            t = []
            for j in range(3):
                t += tri(i + j)
            return t[:8]
        leaf = lambda t: {"search": {"path": "body", "terms": [t]}}
        if workload == "or8":
            return [{"search_req": {"or": {"queries": [leaf(t) for t in terms8(i)]}}, "top": 10} for i in range(batch)]
        return [{"search_req": {"and": {"queries": [{"or": {"queries": [leaf(t) for t in terms8(i)[:4]]}}, {"or": {"queries": [leaf(t) for t in terms8(i)[4:]]}}]}},
                 "top": 10} for i in range(batch)]

    def and_of_ors(i):
        a, b = tri(i), tri(i + 1)
        return synth.req_and_of_ors([a[0], a[1]], [a[2], b[2]], top=10)
    if workload == "config3":
        return [synth.req_and_phrase_locality(tri(i), top=10) for i in range(batch)]
    if workload == "and_of_ors":
        return [and_of_ors(i) for i in range(batch)]
    if workload == "mix":
        return [synth.req_and(tri(i), top=10) if i % 5 in (0, 1) else synth.req_or(tri(i), top=10) if i % 5 in (2, 3) else and_of_ors(i) for i in range(batch)]
    raise ValueError(workload)


def spec_for(workload, docs, terms, triples):
    from veloci_amd import synth
    rich = workload in ("config3", "and_of_ors", "mix")  # these need the phrase pairs, token->text lists and the boost column
    fuzzy = workload == "config4"
    extra = (1000, 100_000, min(1_000_000, docs)) if workload == "config2" else ()
    return synth.SynthSpec(num_docs=docs, num_terms=terms, triples=triples, with_t2t=rich, with_facets=fuzzy, with_boost=rich, with_phrase=rich,
                           background_terms=2000 if fuzzy else 0, extra_probe_dfs=extra)


def kernel_table(prof):
    """vq_profile_json -> {kernel: per-launch time, layout-true GB/s and fraction of the HBM peak, the 8(d) figure beside it}"""
    out = {}
    for name, k in prof.get("kernels", {}).items():
        n = max(k["launches"], 1)
        ms = k["ms"] / n
        lay, alg = k["layout_bytes"] / n, k["algorithmic_bytes"] / n
        gbps = lay / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        agbps = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        gath = k.get("gathered_bytes", 0) / n
        out[name] = {"launch_ms": round(ms, 4), "launches": k["launches"], "queries_per_launch": round(k["queries"] / n, 1), "scan": k["scan"],
                     "layout_bytes_per_launch": int(lay), "gathered_bytes_per_launch": int(gath), "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                     "algorithmic_equiv_GBps": round(agbps, 1)}
    return out


def dominant(table):
    best = None
    for name, k in table.items():
        if best is None or k["launch_ms"] * k["launches"] > table[best]["launch_ms"] * table[best]["launches"]:
            best = name
    return best


class Bench:
    """One staged index (shard) and the searchers over it."""

    def __init__(self, args, workload, docs, terms, triples, rank, world, local_rank, dist_on):
        import veloci_amd
        from veloci_amd import dist as vdist
        from veloci_amd import synth
        self.args, self.rank, self.world, self.dist_on = args, rank, world, dist_on
        t0 = time.time()
        self.lo, self.hi = vdist.shard_range(docs, rank, world)
        self.spec = spec_for(workload, docs, terms, triples)
        self.data, self.meta = synth.generate(self.spec, doc_lo=self.lo, doc_hi=self.hi, device=f"cuda:{local_rank}")
        t_gen = time.time() - t0
        if dist_on:
            vdist.all_reduce_global_lens(self.data)
        t0 = time.time()
        self.index = veloci_amd.Index(self.data, device=local_rank, doc_lo=self.lo, doc_hi=self.hi)
        postings = int(self.data.token_to_anchor_score["body.textindex.to_anchor_id_score"][0][-1])
        if rank == 0:
            log(f"{workload}: docs={docs} shard=[{self.lo},{self.hi}) postings/shard={postings} gen={t_gen:.1f}s load={time.time() - t0:.1f}s "
                f"hbm={self.index.device_bytes / 1e9:.2f} GB")
        self.searcher = vdist.ShardedSearcher(self.index, always_collective=True) if dist_on else None
        self.docs = docs

    def run(self, workload, batch_size, steps, warmup, latency=True, chunks=None, tri_limit=None):
        """-> (qps, ms_per_step, p50 single-query latency ms, kernel table, first hit counts, requests as dicts)"""
        import numpy as np
        import torch
        import veloci_amd
        reqs_json = make_requests(workload, self.meta, batch_size, self.args.probes, tri_limit)
        reqs = [veloci_amd.Request(r) for r in reqs_json]
        batch = veloci_amd.RequestBatch(reqs)
        index, searcher = self.index, self.searcher

        def step():
            # flat C-ABI entry points: no per-result Python objects inside the timed region
            if searcher is not None:
                num_hits, counts, ids, scores, status = searcher.search_batch_flat(batch, stride=10, chunks=chunks)
            else:
                num_hits, counts, ids, scores, status = veloci_amd.search_batch_flat(batch, index, stride=10)
            assert not status.any(), f"request failed: status {status[status != 0][:4]}"
            return num_hits, counts, ids, scores

        def sync():
            torch.cuda.synchronize()
            if self.dist_on:
                import torch.distributed as dist
                dist.barrier()
                torch.cuda.synchronize()

        native = searcher is not None and getattr(searcher, "native", False)
        from veloci_amd import dist as vdist
        step_begin = searcher.step_begin if native else (lambda b: vdist.shard_step_begin(index, b))
        step_end = searcher.step_end if native else vdist.shard_step_end
        # (VQ_BENCH_SYNC_STEPS=1: one vq_search_batch_flat call per step.  Requests with pre-passes — config #4's dictionary scans and unions — stay on
        #  that entry point: it runs such batches on two host threads, which hides more than a second step in flight does: 72 k against 53 k requests/s)
        pipelined = native or (searcher is None and workload != "config4" and os.environ.get("VQ_BENCH_SYNC_STEPS") != "1")

        def run_steps(k):
            """k whole steps -> the outputs of the last one.  Through vq_shard_step_begin / _end (the in-library sharded step; on one GPU the same
            pipeline without an exchange) two steps are kept in flight: step i + 1 is compiled and its scans are queued before step i's
            exchange and merge are waited for."""
            if not pipelined:
                out = None
                for _ in range(k):
                    out = step()
                return out
            out, pending = None, None
            for _ in range(k):
                h = step_begin(batch)
                if pending is not None:
                    out = step_end(pending, 10)
                pending = h
            if pending is not None:
                out = step_end(pending, 10)
            if out is not None:
                assert not out[4].any(), f"request failed: status {out[4][out[4] != 0][:4]}"
                out = out[:4]
            return out

        last = run_steps(warmup)
        index.profile_enable(os.environ.get("VQ_NO_PROFILE") != "1")  # (VQ_NO_PROFILE=1: what the event brackets and counters cost)
        index.profile_json(reset=True)
        sync()
        t0 = time.perf_counter()
        last = run_steps(steps) or last
        sync()
        dt = time.perf_counter() - t0
        prof = index.profile_json(reset=True)
        index.profile_enable(False)
        if self.dist_on:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if self.args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        lat = []
        for i in range(50 if latency else 0):  # single-query latency (p50) through the same path
            a = time.perf_counter()
            if searcher is not None:
                searcher.search_batch([reqs[i % len(reqs)]])
            else:
                veloci_amd.search_batch([reqs[i % len(reqs)]], index)
            lat.append((time.perf_counter() - a) * 1e3)
        p50 = float(np.median(lat[10:])) if lat else None
        self.speculative_reruns = index.speculative_reruns  # (ORs on k_scan_probe_or that had to be run again on the exact kernels, whole life of the index)
        self.last_outputs = tuple(np.array(x) for x in last)  # (num_hits, counts, ids, scores) of the LAST TIMED step: what main() diffs against the oracle
        return batch_size * steps / dt, dt / steps * 1e3, p50, kernel_table(prof), [int(x) for x in last[0][:3]], reqs_json


def roofline_object(table, docs, triples, batch, workload, world):
    if not table:
        return {"bound": "hbm", "note": "profiling was switched off (VQ_NO_PROFILE=1)"}
    name = dominant({k: v for k, v in table.items() if v["scan"]} or table)
    k = table[name]
    roof = {"bound": "hbm", "kernel": name, "achieved": k["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k["frac"],
            "peak_measured_copy": HBM_COPY_GBPS, "frac_of_measured_copy": round(k["GBps"] / HBM_COPY_GBPS, 4), "traffic": None,
            "bytes_min_this_layout_per_launch": k["layout_bytes_per_launch"], "launch_ms": k["launch_ms"], "launches": k["launches"],
            "queries_per_launch": k["queries_per_launch"],
            # SURVEY.md 8(d)'s accounting (6 B per posting of every list + 8 B per returned hit, as if every posting were streamed) beside the
            # bytes this layout really has to move: it exceeds the peak because dense lists are read as bitmap words / 16-bit arrays
            "algorithmic_equiv": {"GBps": k["algorithmic_equiv_GBps"], "frac": round(k["algorithmic_equiv_GBps"] / HBM_PEAK_GBPS, 4)},
            "achieved_is": "bytes this layout must move per launch (DESIGN.md 5) / mean launch time from HIP events on the launch stream inside the timed region"}
    # HBM traffic comes from PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs of this same command under rocprofv3: the counters cannot be read
    # from inside the process): the committed figure is reported when it was taken on exactly this configuration, with its source
    for fn in ("r04_traffic.json", "r03_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                tr = json.load(f)
            c = tr["config"]
            same_launch = c.get("queries_per_launch") is None or abs(float(c["queries_per_launch"]) - float(roof["queries_per_launch"])) < 0.5
            if (c["docs"], c["triples"], c["batch"], c["workload"], c["n_gpus"]) == (docs, triples, batch, workload, world) and same_launch and tr.get("kernel", name) == name:
                roof["traffic"] = tr["traffic_bytes_per_launch"]
                roof["traffic_source"] = f"profiles/{fn}: " + tr.get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected as MI355X_MICROARCH.md prescribes")
                roof["traffic_over_layout_min"] = round(tr["traffic_bytes_per_launch"] / max(k["layout_bytes_per_launch"], 1), 3)
                break
        except (OSError, KeyError, ValueError):
            pass
    return roof


def leg_summary(qps, ms, table):
    """one workload in a few numbers: what the driver's record should still show when the kernel tables are cut off"""
    scans = {k: v for k, v in table.items() if v["scan"]} or table
    name = dominant(scans) if scans else None
    k = table.get(name, {})
    return {"qps": round(qps, 1), "ms_per_step": round(ms, 3), "kernel": name, "launch_ms": k.get("launch_ms"), "frac": k.get("frac"), "queries_per_launch": k.get("queries_per_launch")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=100_000_000)
    ap.add_argument("--triples", type=int, default=DEFAULT_TRIPLES, help="distinct (a,b,c) probe triples; the query stream cycles over them.  A step is ONE launch of --batch queries: with triples >= batch no posting list is read twice inside a launch, so nothing is served from L2 / Infinity Cache that a stream of distinct queries would not find there")
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH)
    ap.add_argument("--terms", type=int, default=100_000, help="dictionary size")
    ap.add_argument("--probes", type=int, default=1024, help="config4: distinct fuzzy probe terms per batch")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--workload", default="and", choices=sorted(WORKLOADS),
                    help="and = the headline metric; config2 / config3 / config4 = BASELINE configs (config4: use --docs 10000000 --terms 1000000); the rest are extra shapes")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the last timed step's outputs (profiling runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of BASELINE configs #2-#4 behind the headline (N=1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-query latency loop (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # Fewer distinct triples than queries per step: the HEADLINE's steps are cut into two launches (VQ_SHARD_CHUNKS, read by the library at every
    # step), so that no launch holds a list twice as long as triples >= batch / 2 — on shards of 40 M docs and more.  Below that (a rank's share
    # at N >= 4) a step stays ONE launch: two launches of 512 queries fill the chip worse there (1/8 shard: 1.67 against 1.37 ms per step), and a
    # list's two readers are 512 queries — more than a gigabyte of other lists — apart, far beyond the 256 MiB the caches hold.
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    launches_per_step = 1
    explicit_chunks = os.environ.get("VQ_SHARD_CHUNKS")  # (a caller's own setting wins, for every leg)
    if explicit_chunks:
        launches_per_step = 2 if explicit_chunks == "2" and args.batch >= 512 else 1
    elif args.triples < args.batch and args.batch >= 512 and os.environ.get("VQ_BENCH_ONE_LAUNCH") != "1" and args.docs // max(world_env, 1) >= 40_000_000 and args.workload == "and":
        launches_per_step = 2
    os.environ["VQ_SHARD_CHUNKS"] = str(launches_per_step)

    # stdout carries ONE line — the result; whatever a library prints there (RCCL greets on stdout when a communicator is made) goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np  # noqa: F401
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # VQ_BENCH_COLLECTIVE=1: take the N>1 code path (process group, sharded searcher, collectives) with however many ranks there are —
    # with one rank this rehearses the RCCL calls of the multi-GPU run on a 1-GPU box
    dist_on = world > 1 or os.environ.get("VQ_BENCH_COLLECTIVE") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import faulthandler
    faulthandler.enable()
    bench = Bench(args, args.workload, args.docs, args.terms, args.triples, rank, world, local_rank, dist_on)
    if rank == 0:
        log("index staged; running", args.steps, "steps of", args.batch, "queries")
    bench_chunks = int(os.environ["VQ_BENCH_CHUNKS"]) if os.environ.get("VQ_BENCH_CHUNKS") else None  # None: the searcher's default
    qps, ms_step, p50, table, first_hits, reqs_json = bench.run(args.workload, args.batch, args.steps, args.warmup, latency=not args.no_latency, chunks=bench_chunks)

    out = None
    if rank == 0:
        log(f"timed region done: {qps:.1f} q/s, {ms_step:.3f} ms per step")
        spec = bench.spec
        out = {
            "metric": "queries/sec, 3-term AND on 100M-doc index (p50 latency and HBM fraction alongside)",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 doc ids + f16->f32 scores", "data": "synthetic",
            "config": {"workload": f"{args.docs}-doc synthetic index, {WORKLOADS[args.workload]} (df 10%/3%/1% of docs, planted overlap), top 10, batches of {args.batch}",
                       "docs": args.docs, "triples": args.triples, "batch": args.batch, "postings_per_query": int(sum(spec.fractions) * args.docs),
                       "launches_per_step": launches_per_step, "distinct_triples_per_launch": min(args.triples, args.batch // launches_per_step),
                       "copies_of_a_list_per_launch": max(1, -(-(args.batch // launches_per_step) // max(args.triples, 1))),
                       "sharding": f"doc-range x{world}", "first_hit_counts": first_hits},
            "p50_latency_ms_single_query": (round(p50, 3) if p50 is not None else None),
            **({"speculative_reruns": bench.speculative_reruns} if args.workload in ("or", "mix") else {}),
            "roofline": roofline_object(table, args.docs, args.triples, args.batch, args.workload, world),
        }
        legs = {"headline": leg_summary(qps, ms_step, table)}
        tables = {"headline": table}
        want_cpu = not args.no_cpu and args.cpu_seconds > 0
        if world == 1 and not dist_on and args.workload in ("and", "or", "single") and (want_cpu or not args.no_parity):
            n_tri = min(2, len(bench.meta.triples))
            t0 = time.time()
            ora = sample_oracle(bench.data, bench.meta, n_tri)
            if not args.no_parity:
                # the timed path checks itself: the last timed step's rows of the first triples against the CPU oracle (outside the timed region)
                out["parity_checked"] = check_timed_outputs(ora, bench.last_outputs, reqs_json, bench.meta, n_tri)
                log(f"parity: {out['parity_checked']} rows of the last timed step equal the oracle's (hit counts, ids, score bits); oracle setup + check {time.time() - t0:.1f}s")
            if want_cpu:
                out["cpu_baseline"] = cpu_baseline(ora, bench.data, bench.meta, reqs_json, args, n_tri)
            del ora

    # ---- the other named shapes, a few steps each (N=1), same accounting: the headline index with launches that hold every list several
    # times (what round 3 reported), BASELINE configs #2 - #4
    if rank == 0 and world == 1 and not dist_on and not args.no_extra and args.workload == "and":

        def short(name, b, workload, steps=8, note=None, launches=1, **kw):
            if not explicit_chunks:
                os.environ["VQ_SHARD_CHUNKS"] = str(launches)  # (the other legs' queries are distinct inside a launch of the whole step)
            try:
                q, ms, p, tab, fh, _ = b.run(workload, args.batch, steps, 3, **kw)
                legs[name] = leg_summary(q, ms, tab)
                legs[name]["index"] = f"{b.docs} docs, {len(b.meta.triples)} triples"
                if p is not None:
                    legs[name]["p50_ms_single_query"] = round(p, 3)
                if note:
                    legs[name]["note"] = note
                tables[name] = tab
            except Exception as ex:  # noqa: BLE001 — an extra line must never take the headline down
                legs[name] = {"error": repr(ex)[:300]}

        if args.triples * launches_per_step >= args.batch and args.batch // launches_per_step >= 8:  # round 3's headline configuration: every list four times inside a launch (256 triples, launches of 1024)
            short("headline_4_copies_per_launch", bench, "and", latency=False, launches=launches_per_step, tri_limit=max(1, args.batch // launches_per_step // 4), note="every list is read by four queries of a launch (round 3's configuration: 256 triples, launches of 1024): a quarter of the footprint")
        short("or_100m_docs", bench, "or", latency=False, note="3-term OR over the headline's lists, one launch per step (k_scan_probe_or; ranks only the docs that hold the rarest term, confirmed by the k-th key)")
        short("config2_100m_docs", bench, "single", latency=False)
        del bench
        import gc
        gc.collect()
        for name, workload, docs, terms, triples, note in (
                ("config2_1m_docs", "config2", 1_000_000, 100_000, 1, "cache-resident: the whole index is 10 MB; this leg measures the host side of a step, its frac is not an HBM figure"),
                ("config3_10m_docs", "config3", 10_000_000, 1_000_000, args.batch, None),
                ("config4_10m_docs", "config4", 10_000_000, 1_000_000, 32, f"{min(args.probes, args.batch)} distinct fuzzy probe terms per batch (what a request reads follows its probe, not the triples)")):
            try:
                b = Bench(args, workload, docs, terms, triples, 0, 1, local_rank, False)
            except Exception as ex:  # noqa: BLE001
                legs[name] = {"error": repr(ex)[:300]}
                continue
            short(name, b, workload, note=note)
            del b
            gc.collect()

    if rank == 0:
        # the compact per-workload summary rides inside `config` (a key the driver's record keeps whole) and once more at the very end of the
        # line (what a tail of the output still shows); the full per-kernel tables sit in between
        out["config"]["legs"] = legs
        out["kernels"] = tables.pop("headline")
        if tables:
            out["configs_kernels"] = tables
        out["summary"] = {"value_qps": out["value"], "frac": out["roofline"].get("frac"), "launch_ms": out["roofline"].get("launch_ms"),
                          "parity_checked": out.get("parity_checked"), "legs": {k: ({"qps": v.get("qps"), "kernel": v.get("kernel"), "frac": v.get("frac")} if "error" not in v else v) for k, v in legs.items()}}

    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def sample_oracle(data, meta, n_tri):
    """The CPU oracle over the posting lists of the first `n_tri` probe triples (the full lists of the index, nothing else)."""
    import numpy as np
    from oracle import binding as O
    from veloci_amd.index import IndexData
    path = "body.textindex.to_anchor_id_score"
    offsets, anchors, scores, _ = data.token_to_anchor_score[path]
    keep = set()
    for tri in meta.triples[:n_tri]:
        for t in tri:
            keep.add(data.term_id("body.textindex", t))
    lens = np.zeros(len(offsets) - 1, np.uint64)
    for t in keep:
        lens[t] = offsets[t + 1] - offsets[t]
    so = np.zeros(len(offsets), np.uint64)
    so[1:] = np.cumsum(lens)
    sa = np.concatenate([anchors[int(offsets[t]):int(offsets[t + 1])] for t in sorted(keep)])
    ss = np.concatenate([scores[int(offsets[t]):int(offsets[t + 1])] for t in sorted(keep)])
    sample = IndexData(data.num_anchors)
    sample.fst = data.fst
    sample.columns = data.columns
    sample.add_token_to_anchor_score(path, so, sa, ss)
    ora = O.OracleIndex(data.num_anchors)
    sample.load_into(ora)
    return ora


def check_timed_outputs(ora, outputs, reqs_json, meta, n_tri, max_distinct=8):
    """The rows of the LAST TIMED step whose terms all lie in the sample's triples against the oracle: hit counts, doc ids and score bits.
    -> number of rows checked; raises SystemExit(1) on the first difference (a fast wrong answer is not a result)."""
    import numpy as np
    num_hits, counts, ids, scores = outputs
    kept = {t for tri in meta.triples[:n_tri] for t in tri}

    def terms_of(node, out):
        if isinstance(node, dict):
            for k, v in node.items():
                if k == "terms":
                    out.update(v)
                else:
                    terms_of(v, out)
        elif isinstance(node, list):
            for v in node:
                terms_of(v, out)
        return out

    want = {}
    checked = 0
    for i, r in enumerate(reqs_json):
        if not terms_of(r, set()) <= kept:
            continue
        key = json.dumps(r, sort_keys=True)
        if key not in want:
            if len(want) >= max_distinct:
                continue
            want[key] = ora.search_json(json.dumps(r))
        w = want[key]
        c = int(counts[i])
        ok = int(num_hits[i]) == w.num_hits and c == len(w.ids) and np.array_equal(ids[i, :c], np.asarray(w.ids, np.uint32)) and \
            np.array_equal(np.asarray(scores[i, :c], np.float32).view(np.uint32), np.asarray(w.scores, np.float32).view(np.uint32))
        if not ok:
            log(f"PARITY FAILURE: row {i} of the last timed step differs from the oracle: {json.dumps(r)}\n  got  {int(num_hits[i])} hits {ids[i, :c].tolist()} {scores[i, :c].tolist()}"
                f"\n  want {w.num_hits} hits {list(w.ids)} {list(w.scores)}")
            raise SystemExit(1)
        checked += 1
    return checked


def cpu_baseline(ora, data, meta, reqs_json, args, n_tri):
    """CPU oracle on a bounded sample: the first probe triples' posting lists, the same AND requests; >= 1000 measured queries in the
    all-core mode (one independent query per thread: how the reference is driven by its server threads), single-thread p50 beside it."""
    import numpy as np
    t0 = time.time()
    sample_reqs = [json.dumps(r) for r in reqs_json[:n_tri]]
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    # single thread: a few queries for the latency (the budget goes to the throughput run)
    secs, lat, _ = ora.bench(sample_reqs, repeat=1, threads=1)
    per_q = secs / len(sample_reqs)
    rep1 = max(1, int(args.cpu_seconds * 0.2 / max(per_q, 1e-6) / len(sample_reqs)))
    secs1, lat1, _ = ora.bench(sample_reqs, repeat=rep1, threads=1)
    # one independent query per thread, on 16 threads (one GPU's share of this pool's hosts), 64, and all visible cores (at most 128: every
    # thread materialises its operands' hit lists, ~250 MB): the best of them is the baseline (VQ_CPU_THREADS pins one count)
    if os.environ.get("VQ_CPU_THREADS"):
        candidates = [max(1, min(visible, int(os.environ["VQ_CPU_THREADS"])))]
    else:
        candidates = sorted({max(1, min(visible, c)) for c in (16, 64, min(visible, 128))})
    sweep = {}
    for c in candidates:
        repc = max(1, (2 * c) // len(sample_reqs))  # two queries per thread: the second runs on warm arenas
        secsc, _, _ = ora.bench(sample_reqs, repeat=repc, threads=c)
        sweep[c] = len(sample_reqs) * repc / secsc
    cores = max(sweep, key=sweep.get)
    # every visible core as a figure of its own (when the host's free memory holds one query's hit lists per thread, ~0.4 GB each)
    all_cores_qps, all_cores_note = sweep.get(visible), None
    if all_cores_qps is None:
        try:
            with open("/proc/meminfo") as f:
                avail_gb = next(int(line.split()[1]) for line in f if line.startswith("MemAvailable")) / 1e6
        except (OSError, StopIteration, ValueError):
            avail_gb = 0.0
        if avail_gb > 0.4 * visible + 8:
            secsv, _, _ = ora.bench(sample_reqs, repeat=max(1, visible // len(sample_reqs)), threads=visible)
            all_cores_qps = len(sample_reqs) * max(1, visible // len(sample_reqs)) / secsv
            sweep[visible] = all_cores_qps
        else:
            all_cores_note = f"not run: {avail_gb:.0f} GB available for {visible} threads x ~0.4 GB of hit lists"
    per_rep = len(sample_reqs) / sweep[cores]
    repn = max(int(np.ceil(1000 / len(sample_reqs))), int(args.cpu_seconds * 0.6 / max(per_rep, 1e-6)))
    if repn * per_rep > 3 * args.cpu_seconds:  # a slow host: stay bounded, say so in `sample`
        repn = max(1, int(3 * args.cpu_seconds / max(per_rep, 1e-6)))
    secsn, latn, _ = ora.bench(sample_reqs, repeat=repn, threads=cores)
    qps_n = len(sample_reqs) * repn / secsn
    log(f"cpu baseline: setup {time.time() - t0:.1f}s, 1 thread {len(sample_reqs) * rep1 / secs1:.2f} q/s, {cores} threads {qps_n:.2f} q/s over {len(sample_reqs) * repn} queries")
    return {"value": round(qps_n, 3), "unit": "queries/s", "cores": cores, "cores_visible": visible, "kind": "port", "threads_tried_qps": {str(c): round(v, 2) for c, v in sweep.items()},
            "all_visible_cores_qps": (round(all_cores_qps, 2) if all_cores_qps is not None else None), **({"all_visible_cores_note": all_cores_note} if all_cores_note else {}),
            "measured_queries": len(sample_reqs) * repn, "p50_ms_all_cores": round(float(np.median(latn)) / 1e6, 3), "p95_ms_all_cores": round(float(np.percentile(latn, 95)) / 1e6, 3),
            "single_thread_qps": round(len(sample_reqs) * rep1 / secs1, 3), "single_thread_p50_ms": round(float(np.median(lat1)) / 1e6, 3),
            "sample": f"C++ restatement of the reference algorithm (oracle/, not the Rust binary); {n_tri} of the {len(meta.triples)} probe triples, "
                      f"{len(sample_reqs) * repn} 3-term AND queries on the full {data.num_anchors}-doc lists with one independent query per thread on {cores} threads "
                      f"(+ {len(sample_reqs) * rep1} on one thread for the latency)"}


if __name__ == "__main__":
    main()
