# k_scan_probe against k_scan_simple on one doc-range shard (a rank's share at N = 8 / 4).
# usage: bash tools/probe_shard_sweep.sh  (on the GPU box)
run() {
  echo "== docs=$1 env=$2"
  env VQ_BENCH_COLLECTIVE=1 VQ_BENCH_ONE_LAUNCH=1 $2 timeout -k 10 300 python -u bench.py --docs $1 --triples 256 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>gpurun_out/sw.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['frac'], d.get('parity_checked'))"
  grep -E "Traceback|Error" gpurun_out/sw.err | head -3
}
for docs in 12500000 25000000; do
  run $docs "VQ_PROBE_MIN_DOCS=40000000"
  run $docs "VQ_PROBE_MIN_DOCS=0"
done
