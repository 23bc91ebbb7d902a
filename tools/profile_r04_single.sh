# kernel stats + FETCH_SIZE of the single-term scan (k_scan_union on the tile-packed words) — usage: bash tools/profile_r04_single.sh  (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_single; mkdir -p $O
B="bench.py --workload single --docs 100000000 --triples 256 --steps 4 --warmup 2 --no-cpu --no-extra --no-latency"
S=/tmp/r04_single_stats; rm -rf $S
VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 $B > $O/bench_stats.json 2> $O/bench_stats.err || exit 1
f=$(find $S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep "vq::" $f) > $O/kernel_stats.csv
S=/tmp/r04_single_fetch; rm -rf $S
VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc FETCH_SIZE -d $S -o p -- python3 $B > $O/bench_fetch.json 2> $O/bench_fetch.err || exit 1
python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1; (head -1 $S/sum.csv; grep "vq::" $S/sum.csv) > $O/pmc_fetch.csv
cat $O/kernel_stats.csv | cut -c1-60,150-; cat $O/pmc_fetch.csv
python3 -c "
import json
for n in ('stats','fetch'):
    d=json.loads(open('$O/bench_'+n+'.json').read().strip().splitlines()[-1]); r=d['roofline']; print(n, d['value'], r['kernel'], r['launch_ms'], r['frac'], r['bytes_min_this_layout_per_launch'])"
