# kernel stats, FETCH_SIZE and SQ instruction counts of the 3-term OR on 100 M docs (k_scan_probe_or) — usage: bash tools/profile_r04_or.sh  (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_or; mkdir -p $O
B="bench.py --workload or --docs 100000000 --triples 256 --steps 4 --warmup 2 --no-cpu --no-extra --no-latency"
S=/tmp/r04_or_stats; rm -rf $S
VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 $B > $O/bench_stats.json 2> $O/bench_stats.err || exit 1
f=$(find $S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep "vq::" $f) > $O/kernel_stats.csv
for pass in "fetch FETCH_SIZE" "sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES"; do
  set -- $pass; name=$1; shift
  S=/tmp/r04_or_$name; rm -rf $S
  VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc "$@" -d $S -o p -- python3 $B > $O/bench_$name.json 2> $O/bench_$name.err || exit 1
  python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1; (head -1 $S/sum.csv; grep "vq::k_scan" $S/sum.csv) > $O/pmc_$name.csv
done
cut -c1-40,140- $O/kernel_stats.csv | head -3; cat $O/pmc_fetch.csv $O/pmc_sq_insts.csv
python3 -c "
import json
d=json.loads(open('$O/bench_stats.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['kernel'], r['launch_ms'], r['frac'], r['bytes_min_this_layout_per_launch'], d.get('parity_checked'), d.get('speculative_reruns'))"
