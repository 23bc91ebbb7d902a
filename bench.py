#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): queries/sec of 3-term AND on a 100 M-doc synthetic index.

    python bench.py --gpus N --steps K --warmup W

A step is one batch of --batch (default 1024) 3-term AND queries, top 10, executed through the C ABI
(`vq_search_batch` / the sharded partial + merge path).  The index is fixed at --docs (default 1e8)
documents and sharded by doc-id range over the N ranks (strong scaling); per-shard top-k are merged
after one RCCL all-gather per batch.  Rank 0 prints ONE JSON line.

roofline   : dominant kernel k_scan_simple (3-term AND; other workloads name theirs); achieved = algorithmic bytes (6 B per posting of the three
             lists + 8 B per returned hit) per launch / mean launch time measured with HIP
             events on the launch stream inside the library (vq_profile_read).
cpu_baseline: the CPU oracle (C++ restatement of the reference algorithm, `kind: port`) timed on this
             host on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=100_000_000)
    ap.add_argument("--triples", type=int, default=32, help="distinct (a,b,c) probe triples; the query stream cycles over them")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--terms", type=int, default=100_000, help="dictionary size")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--workload", default="and", choices=["and", "or", "single", "config3", "and_of_ors", "mix", "config4", "or8", "and_of_or4"],
                    help="and = the headline metric; or / single = extra shapes; config3 = AND + 2 phrase pairs + text locality; "
                         "and_of_ors = AND(OR,OR) + Log10 boost + phrase + locality; mix = 40%% and / 40%% or / 20%% and_of_ors (BASELINE configs #3 / #5); "
                         "config4 = lev-2 fuzzy term + facets on cat and tags[] (use --docs 10000000 --terms 1000000)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-query latency loop (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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

    import veloci_amd
    from veloci_amd import dist as vdist
    from veloci_amd import synth

    # ---- synthetic index shard (deterministic; shard == slice of the unsharded index)
    t0 = time.time()
    lo, hi = vdist.shard_range(args.docs, rank, world)
    rich = args.workload in ("config3", "and_of_ors", "mix")  # these need the phrase pairs, token->text lists and the boost column
    fuzzy = args.workload == "config4"
    spec = synth.SynthSpec(num_docs=args.docs, num_terms=args.terms, triples=args.triples, with_t2t=rich, with_facets=fuzzy, with_boost=rich,
                           with_phrase=rich, background_terms=2000 if fuzzy else 0)
    data, meta = synth.generate(spec, doc_lo=lo, doc_hi=hi, device=f"cuda:{local_rank}")
    t_gen = time.time() - t0
    if dist_on:
        vdist.all_reduce_global_lens(data)
    t0 = time.time()
    index = veloci_amd.Index(data, device=local_rank, doc_lo=lo, doc_hi=hi)
    t_load = time.time() - t0
    postings = int(data.token_to_anchor_score["body.textindex.to_anchor_id_score"][0][-1])
    if rank == 0:
        log(f"docs={args.docs} shard=[{lo},{hi}) postings/shard={postings} gen={t_gen:.1f}s load={t_load:.1f}s hbm={index.device_bytes / 1e9:.2f} GB")

    if args.workload == "and":
        reqs_json = [synth.req_and(list(meta.triples[i % len(meta.triples)]), top=10) for i in range(args.batch)]
    elif args.workload == "or":
        reqs_json = [synth.req_or(list(meta.triples[i % len(meta.triples)]), top=10) for i in range(args.batch)]
    elif args.workload == "config4":
        # 100 distinct query terms: vocabulary terms with posting lists, 1-2 random edits each
        rng = np.random.default_rng(4)
        pool = [t for tri in meta.triples for t in tri] + list(meta.background)
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
        qterms = [edit(pool[int(rng.integers(0, len(pool)))]) for _ in range(100)]
        reqs_json = [{"search_req": {"search": {"path": "body", "terms": [qterms[i % len(qterms)]], "levenshtein_distance": 2}}, "top": 10,
                      "facets": [{"field": "cat"}, {"field": "tags[]"}]} for i in range(args.batch)]
    elif args.workload in ("or8", "and_of_or4"):
        # shapes of the reference's query generator (one term expanded over several fields, src/query_generator.rs): a flat OR over 8
        # leaves, and an AND of two 4-leaf ORs
        def terms8(i):
            t = []
            for j in range(3):
                t += list(meta.triples[(i + j) % len(meta.triples)])
            return t[:8]
        leaf = lambda t: {"search": {"path": "body", "terms": [t]}}
        if args.workload == "or8":
            reqs_json = [{"search_req": {"or": {"queries": [leaf(t) for t in terms8(i)]}}, "top": 10} for i in range(args.batch)]
        else:
            reqs_json = [{"search_req": {"and": {"queries": [{"or": {"queries": [leaf(t) for t in terms8(i)[:4]]}}, {"or": {"queries": [leaf(t) for t in terms8(i)[4:]]}}]}},
                          "top": 10} for i in range(args.batch)]
    elif args.workload == "single":
        reqs_json = [synth.req_single(meta.triples[i % len(meta.triples)][i // len(meta.triples) % 3], top=10) for i in range(args.batch)]
    else:
        tri = lambda i: list(meta.triples[i % len(meta.triples)])
        def and_of_ors(i):
            a, b = tri(i), tri(i + 1)
            return synth.req_and_of_ors([a[0], a[1]], [a[2], b[2]], top=10)
        if args.workload == "config3":
            reqs_json = [synth.req_and_phrase_locality(tri(i), top=10) for i in range(args.batch)]
        elif args.workload == "and_of_ors":
            reqs_json = [and_of_ors(i) for i in range(args.batch)]
        else:
            reqs_json = [synth.req_and(tri(i), top=10) if i % 5 in (0, 1) else synth.req_or(tri(i), top=10) if i % 5 in (2, 3) else and_of_ors(i)
                         for i in range(args.batch)]
    reqs = [veloci_amd.Request(r) for r in reqs_json]
    batch = veloci_amd.RequestBatch(reqs)
    searcher = vdist.ShardedSearcher(index, always_collective=True) if dist_on else None
    bench_chunks = int(os.environ["VQ_BENCH_CHUNKS"]) if os.environ.get("VQ_BENCH_CHUNKS") else None  # None: the searcher's default

    class Row:  # what the bench looks at of a result
        def __init__(self, nh):
            self.num_hits = int(nh)

    def step():
        # flat C-ABI entry points: no per-result Python objects inside the timed region
        if searcher is not None:
            num_hits, counts, ids, scores, status = searcher.search_batch_flat(batch, stride=10, chunks=bench_chunks)
        else:
            num_hits, counts, ids, scores, status = veloci_amd.search_batch_flat(batch, index, stride=10)
        assert not status.any()
        return [Row(x) for x in num_hits[:3]]

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    index.profile_enable(True)
    index.profile_read(reset=True)
    sync()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        res = step()
        step_times.append(time.perf_counter() - ts)
    t_loop = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    if rank == 0 and os.environ.get("VQ_TIMING"):
        log("step times ms:", [round(x * 1e3, 2) for x in step_times], "loop", round(t_loop * 1e3, 2), "with final sync", round(dt * 1e3, 2))
    scan_ms, launches, algo_bytes = index.profile_read(reset=True)
    index.profile_enable(False)
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_q = args.batch * args.steps
    qps = total_q / dt

    # single-query latency (p50) through the same path
    lat = []
    for i in range(0 if args.no_latency else 50):
        a = time.perf_counter()
        if searcher is not None:
            searcher.search_batch([reqs[i % len(reqs)]])
        else:
            veloci_amd.search_batch([reqs[i % len(reqs)]], index)
        lat.append((time.perf_counter() - a) * 1e3)
    p50 = float(np.median(lat[10:])) if lat else float('nan')

    out = None
    if rank == 0:
        per_launch_ms = scan_ms / max(launches, 1)
        per_launch_bytes = algo_bytes / max(launches, 1)
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        peak = 8000.0
        out = {
            "metric": "queries/sec, 3-term AND on 100M-doc index (p50 latency and HBM fraction alongside)",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 doc ids + f16->f32 scores", "data": "synthetic",
            "config": {"workload": f"{args.docs}-doc synthetic index, " + {"and": "3-term AND", "or": "3-term OR", "single": "single-term scan",
                                                                              "config3": "3-term AND + 2 phrase pairs + text locality",
                                                                              "and_of_ors": "AND(OR,OR) + Log10 boost + phrase + locality",
                                                                              "mix": "40% AND / 40% OR / 20% AND(OR,OR)+boost+phrase+locality",
                                                                              "config4": "lev-2 fuzzy term + facets (cat, tags[])", "or8": "flat OR over 8 terms",
                                                                              "and_of_or4": "AND of two 4-term ORs"}[args.workload] +
                       f" (df 10%/3%/1% of docs, planted overlap), top 10, batches of {args.batch}",
                       "docs": args.docs, "triples": args.triples, "batch": args.batch, "postings_per_query": int(sum(spec.fractions) * args.docs),
                       "sharding": f"doc-range x{world}", "first_hit_counts": [int(r.num_hits) for r in res[:3]]},
            "p50_latency_ms_single_query": (round(p50, 3) if p50 == p50 else None),
            "roofline": {"bound": "hbm", "kernel": {"and": "k_scan_simple", "or": "k_scan_simple", "single": "k_scan_union", "config3": "k_scan_simple<rich>", "and_of_ors": "k_scan_simple<rich>",
                                                    "mix": "k_scan_simple (plain + rich launches)", "config4": "k_scan_simple<rich> (+ k_dict_scan, k_union pre-passes)"}.get(args.workload, "k_tile_scan"), "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
                         "frac": round(achieved / peak, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(per_launch_bytes), "launch_ms": round(per_launch_ms, 4), "launches": int(launches),
                         "note": "achieved = algorithmic bytes (6 B per posting + 8 B per returned hit, SURVEY.md 8d) / mean launch time; the kernel reads dense "
                                 "lists as bitmap images, so it moves fewer HBM bytes than that (see traffic) and frac can exceed 1"},
        }
        # HBM traffic per launch from the committed PMC passes (profiles/r01_traffic.json), when they were taken on this configuration
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                tr = json.load(f)
            c = tr["config"]
            if (c["docs"], c["triples"], c["batch"], c["workload"], c["n_gpus"]) == (args.docs, args.triples, args.batch, args.workload, world):
                out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            pass
        if world == 1 and not args.no_cpu and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(data, meta, reqs_json, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(data, meta, reqs_json, args):
    """CPU oracle on a bounded sample: the first two probe triples' posting lists, the same AND requests."""
    from oracle import binding as O
    from veloci_amd.index import IndexData
    t0 = time.time()
    path = "body.textindex.to_anchor_id_score"
    offsets, anchors, scores, _ = data.token_to_anchor_score[path]
    n_tri = min(2, len(meta.triples))
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
    sample_reqs = [json.dumps(r) for r in reqs_json[:n_tri]]
    # calibrate, then spend about --cpu-seconds in total over the two modes
    secs, lat, _ = ora.bench(sample_reqs, repeat=1, threads=1)
    per_q = secs / len(sample_reqs)
    # host cores of this job's share of the box (one GPU's share is 16 cores on this pool)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    rep1 = max(1, int(args.cpu_seconds * 0.4 / max(per_q, 1e-6) / len(sample_reqs)))
    secs1, lat1, _ = ora.bench(sample_reqs, repeat=rep1, threads=1)
    # all-core mode does not scale linearly (allocation + memory bound): calibrate it on its own
    repc = max(1, cores // len(sample_reqs))
    secsc, _, _ = ora.bench(sample_reqs, repeat=repc, threads=cores)
    repn = max(repc, int(args.cpu_seconds * 0.6 / max(secsc, 1e-6) * repc))
    secsn, latn, _ = ora.bench(sample_reqs, repeat=repn, threads=cores)
    qps_n = len(sample_reqs) * repn / secsn
    log(f"cpu baseline: setup {time.time() - t0:.1f}s, 1 thread {len(sample_reqs) * rep1 / secs1:.2f} q/s, {cores} threads {qps_n:.2f} q/s")
    return {"value": round(qps_n, 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "single_thread_qps": round(len(sample_reqs) * rep1 / secs1, 3), "single_thread_p50_ms": round(float(np.median(lat1)) / 1e6, 3),
            "sample": f"C++ restatement of the reference algorithm (oracle/, not the Rust binary); {n_tri} of the {len(meta.triples)} probe triples, "
                      f"{len(sample_reqs) * (rep1 + repn)} 3-term AND queries on the full {data.num_anchors}-doc lists, one independent query per thread"}


if __name__ == "__main__":
    main()
