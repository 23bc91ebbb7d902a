# GPU-side timeline of the headline's steps: gaps between the vq:: kernels (rocprofv3 --kernel-trace).  usage: bash tools/step_timeline.sh  (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=/tmp/r04_timeline; rm -rf $S
timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extra --no-latency --no-parity > gpurun_out/tl.json 2> gpurun_out/tl.err || exit 1
f=$(find $S -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "vq::" in r["Kernel_Name"]][-40:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{r['Kernel_Name'].split('(')[0][4:]:22s} start +{gap:8.1f} us after the previous end, ran {(e - s) / 1e3:9.1f} us, queue {r.get('Queue_Id', '?')}")
    prev_end = max(prev_end or 0, e)
PY
