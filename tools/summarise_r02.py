"""Turn the raw outputs of tools/profile_r02.sh into the committed evidence (profiles/r02_*): run ON the GPU box by that script (the raw counter tables
are too large to travel), writing into gpurun_out/r02/, from where the summaries are copied to profiles/.
usage: python tools/summarise_r02.py <raw dir> <out dir>"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02")   # raw outputs (on the GPU box: /tmp, they run to hundreds of MB)
DST = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")            # the small summaries


def json_line(path):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def first(pattern):
    g = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    return g[0] if g else None


def counters(pattern):
    """-> {kernel: {counter: mean per dispatch}}, {kernel: dispatches}"""
    f = first(pattern)
    if not f:
        return {}, {}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    n = {k: len(v) for k, v in seen.items()}
    return {k: {c: v / n[k] for c, v in cs.items()} for k, cs in per.items()}, n


def main():
    os.makedirs(DST, exist_ok=True)
    # bench lines
    for src, dst in (("bench_n1.json", "r02_bench_n1.json"), ("bench_under_rocprof.json", "r02_bench_under_rocprof.json"), ("bench_or8.json", "r02_bench_or8_100m.json"),
                     ("bench_and_of_or4.json", "r02_bench_and_of_or4_100m.json"), ("bench_or.json", "r02_bench_or.json"), ("bench_single.json", "r02_bench_single.json"),
                     ("bench_config4_10m_256triples.json", "r02_bench_config4_10m.json"), ("bench_shard8_collective.json", "r02_bench_shard8_collective_path.json"),
                     ("bench_jmdict_shape.json", "r02_bench_jmdict_shape.json"), ("full_vocab_footprint.json", "r02_full_vocabulary_footprint.json")):
        d = json_line(os.path.join(SRC, src))
        if d is not None:
            with open(os.path.join(DST, dst), "w") as f:
                json.dump(d, f, indent=1)
                f.write("\n")
    # rocprofv3 --kernel-trace --stats summary
    st = first("stats/**/*kernel_stats.csv")
    if st:
        shutil.copyfile(st, os.path.join(DST, "r02_kernel_stats.csv"))
    # HBM traffic of the dominant kernel of the headline bench
    fetch, nf = counters("fetch/**/*counter_collection.csv")
    write, nw = counters("write/**/*counter_collection.csv")
    bench = json_line(os.path.join(SRC, "bench_fetch.json")) or json_line(os.path.join(SRC, "bench_under_rocprof.json"))
    key = next((k for k in fetch if "k_scan_simple" in k and "true" not in k), None)
    if key and bench:
        kt = next(v for k, v in bench["kernels"].items() if "k_scan_simple" in k)
        fetch_b = fetch[key]["FETCH_SIZE"] * 1024.0
        write_b = write.get(key, {}).get("WRITE_SIZE", 0.0) * 1024.0
        streamed = kt["layout_bytes_per_launch"] - kt["gathered_bytes_per_launch"]
        gather_reported = max(fetch_b - streamed / 2.0, 0.0)
        out = {
            "_comment": "HBM traffic of one k_scan_simple (AND) launch of the default bench, from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; mean per dispatch). "
                        "Corrected with this repo's own calibration of FETCH_SIZE on gfx950 (profiles/r02_fetch_size_calibration.txt, tools/fetch_calib.hip): a 16 B/lane coalesced stream is "
                        "reported at HALF its bytes (the guide's x2), a lone 2-byte gather at exactly its 64-byte sector, two gathers sharing a 128-byte line at half. The launch's streamed "
                        "bytes are known exactly (bitmap words + doc ids of the layout), so: traffic = streamed + (FETCH - streamed/2) * g + WRITE with g in [1, 2] for the gather sectors; "
                        "traffic_bytes_per_launch uses g = 1 (lone sectors, the common case at ~1 % survivor density), the upper bound g = 2 is given beside it.",
            "config": {"docs": bench["config"]["docs"], "triples": bench["config"]["triples"], "batch": bench["config"]["batch"], "workload": "and", "n_gpus": 1,
                       "queries_per_launch": kt["queries_per_launch"]},
            "dispatches_averaged": nf.get(key),
            "fetch_size_bytes_reported_per_launch": round(fetch_b), "write_size_bytes_per_launch": round(write_b),
            "streamed_bytes_per_launch_layout": int(streamed), "gather_sector_bytes_reported": round(gather_reported),
            "traffic_bytes_per_launch": round(streamed + gather_reported + write_b), "traffic_bytes_per_launch_upper_bound": round(streamed + 2 * gather_reported + write_b),
            "guide_prescribed_2x_fetch_plus_write": round(2 * fetch_b + write_b),
            "launch_ms_under_profiler": kt["launch_ms"],
            "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --no-latency",
        }
        with open(os.path.join(DST, "r02_traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        for name, table in (("r02_pmc_fetch_size_bench.csv", "fetch/**/*counter_collection.csv"), ("r02_pmc_write_size_bench.csv", "write/**/*counter_collection.csv")):
            src = first(table)
            if src:  # keep the dominant kernel's rows only (the full tables run to tens of MB)
                rows = [r for r in csv.DictReader(open(src)) if "k_scan_simple" in r["Kernel_Name"]]
                with open(os.path.join(DST, name), "w", newline="") as f:
                    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                    w.writeheader()
                    w.writerows(rows[:400])
    # SQ counters of the 8-leaf OR: generic interpreter vs the wide kernel
    rep = {}
    for tag, pats in (("k_tile_scan (VQ_NO_WIDE=1)", ("sq_tile/**/*counter_collection.csv", "sq2_tile/**/*counter_collection.csv")),
                      ("k_scan_wide", ("sq_wide/**/*counter_collection.csv", "sq2_wide/**/*counter_collection.csv"))):
        merged = {}
        for pat in pats:
            c, n = counters(pat)
            for k, v in c.items():
                if "k_tile_scan" in k or "k_scan_wide" in k:
                    merged.setdefault(k.split("(")[0], {}).update({cn: round(cv) for cn, cv in v.items()})
                    merged[k.split("(")[0]]["dispatches"] = n[k]
        rep[tag] = merged
    for tag, f in (("k_tile_scan (VQ_NO_WIDE=1)", "or8_tile.json"), ("k_scan_wide", "or8_wide.json")):
        d = json_line(os.path.join(SRC, f))
        if d:
            rep[tag]["bench_under_profiler"] = {"queries_per_s": d["value"], "kernels": {k: {"launch_ms": v["launch_ms"], "queries_per_launch": v["queries_per_launch"]} for k, v in d["kernels"].items() if v["scan"]}}
    if any(rep.values()):
        rep["_about"] = ("Flat OR over 8 terms, 100 M docs, 256-query launches: mean SQ counters per dispatch (rocprofv3 --pmc, two passes of four counters each) of the scan kernel "
                         "that serves the shape — the generic interpreter k_tile_scan (VQ_NO_WIDE=1) before, k_scan_wide after (VERDICT r1 item 4).")
        with open(os.path.join(DST, "r02_pmc_sq_or8_tile_vs_wide.json"), "w") as f:
            json.dump(rep, f, indent=1)
            f.write("\n")
    print("profiles/:", sorted(p for p in os.listdir(DST) if p.startswith("r02_")))


if __name__ == "__main__":
    main()
