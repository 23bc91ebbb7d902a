line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['frac'], d.get('parity_checked'), d['config'].get('launches_per_step'))"
}
echo "== N>1 code path, one rank, 1/8 shard, bench defaults"
VQ_BENCH_COLLECTIVE=1 timeout -k 10 300 python -u bench.py --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err | line; grep -E "Traceback|Error" gpurun_out/e.err | head
echo "== same, one launch per step"
VQ_BENCH_COLLECTIVE=1 VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 300 python -u bench.py --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err | line; grep -E "Traceback|Error" gpurun_out/e.err | head
echo "== mix on the 1/8 shard through the collective path"
VQ_BENCH_COLLECTIVE=1 timeout -k 10 300 python -u bench.py --workload mix --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err | line; grep -E "Traceback|Error" gpurun_out/e.err | head
for fs in 4096 16384 65536; do
  echo "== config4 VQ_FACET_SPAN_POSTINGS=$fs"
  VQ_FACET_SPAN_POSTINGS=$fs timeout -k 10 300 python -u bench.py --workload config4 --docs 10000000 --terms 1000000 --triples 32 --steps 8 --warmup 3 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err | line; grep -E "Traceback|Error" gpurun_out/e.err | head
done
echo "== mix on 100 M docs"
timeout -k 10 400 python -u bench.py --workload mix --docs 100000000 --triples 256 --steps 8 --warmup 3 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err > gpurun_out/mix100.json; cat gpurun_out/mix100.json | line; grep -E "Traceback|Error" gpurun_out/e.err | head
