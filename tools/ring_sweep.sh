#!/bin/bash
# k_scan_ring tuning runs on the GPU box: one bench process per setting (the knobs are read once per process), 64 probe triples
# usage: tools/ring_sweep.sh <tag> "ENV=val ENV2=val" ["ENV=val" ...]
tag=$1; shift
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' =/' '___')
  env $cfg timeout -k 10 300 python bench.py --triples 64 --steps 8 --warmup 3 --no-extra --no-cpu --no-latency ${BENCH_ARGS} > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || { echo "FAILED $cfg"; tail -3 gpurun_out/${tag}_${name}.err; exit 1; }
  python3 - "$cfg" gpurun_out/${tag}_${name}.json <<'P'
import json,sys
d=json.load(open(sys.argv[2]))
k=[(n,v) for n,v in d["kernels"].items() if v["scan"]]
print("%-60s %9.0f q/s  step %.3f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s %.3f ms frac %.3f" % (n, v["launch_ms"], v["frac"]) for n,v in k), "parity", d.get("parity_checked"), flush=True)
P
done
