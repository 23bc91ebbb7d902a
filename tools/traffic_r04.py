#!/usr/bin/env python3
"""HBM traffic of the headline kernel from a tools/profile_r04.sh output directory -> JSON (what bench.py reads as profiles/r04_traffic.json).
FETCH_SIZE on gfx950 reports a 16 B/lane coalesced stream at HALF its bytes and a lone 2-byte gather at its whole 64-byte sector
(MI355X_MICROARCH.md, HBM section; this repo's calibration: profiles/r02_fetch_size_calibration.txt), so with the streamed bytes known from the
layout:  traffic = streamed + (FETCH - streamed / 2) + WRITE;  the guide's blanket rule (2 x FETCH + WRITE) is kept beside it as the upper bound."""
import csv
import json
import sys

d = sys.argv[1]


def mean(name, counter, kernel="k_scan_probe"):
    with open(f"{d}/pmc_{name}.csv", newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                return float(row["Mean_per_dispatch"]), int(row["Dispatches"])
    raise SystemExit(f"{counter} of {kernel} not found in {d}/pmc_{name}.csv")


bench = json.loads(open(f"{d}/bench_fetch.json").read().strip().splitlines()[-1])
roof, cfg = bench["roofline"], bench["config"]
k = bench["kernels"][roof["kernel"]]
fetch, n = mean("fetch", "FETCH_SIZE")   # rocprofv3 reports both in KiB
write, _ = mean("write", "WRITE_SIZE")
fetch, write = fetch * 1024.0, write * 1024.0
layout, gathered = k["layout_bytes_per_launch"], k["gathered_bytes_per_launch"]
streamed = layout - gathered
traffic = streamed + (fetch - streamed / 2) + write
out = {
    "_comment": __doc__.split("\n\n")[0].replace("\n", " ") if False else "HBM traffic of one launch of the headline kernel of the default bench, from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; mean per dispatch), corrected with the gfx950 FETCH_SIZE rule (a 16 B/lane stream is reported at half its bytes, a lone gather at its 64-byte sector): traffic = streamed + (FETCH - streamed/2) + WRITE; the guide's blanket 2 x FETCH + WRITE beside it",
    "kernel": roof["kernel"],
    "config": {"docs": cfg["docs"], "triples": cfg["triples"], "batch": cfg["batch"], "workload": "and", "n_gpus": bench["n_gpus"], "queries_per_launch": roof["queries_per_launch"]},
    "dispatches_averaged": n,
    "fetch_size_bytes_reported_per_launch": int(fetch),
    "write_size_bytes_per_launch": int(write),
    "streamed_bytes_per_launch_layout": int(streamed),
    "gather_sector_bytes_reported": int(fetch - streamed / 2),
    "layout_min_bytes_per_launch": int(layout),
    "traffic_bytes_per_launch": int(traffic),
    "traffic_over_layout_min": round(traffic / layout, 3),
    "guide_prescribed_2x_fetch_plus_write": int(2 * fetch + write),
    "launch_ms_under_profiler": roof["launch_ms"],
    "how": "tools/profile_r04.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-extra --no-latency",
}
try:
    rd, _ = mean("tcc_ea", "TCC_EA0_RDREQ_sum")
    out["tcc_ea0_rdreq_per_launch"] = int(rd)
except SystemExit:
    pass
print(json.dumps(out, indent=1))
