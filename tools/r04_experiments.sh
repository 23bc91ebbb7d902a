# round-4 knob experiments (on the GPU box): config #4 against the facet span length
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['frac']); print({k:(v['launch_ms'],v['queries_per_launch']) for k,v in d['kernels'].items()})"
}
for fs in 512 1024 2048 4096; do
  echo "== config4 VQ_FACET_SPAN_POSTINGS=$fs"
  VQ_FACET_SPAN_POSTINGS=$fs timeout -k 10 300 python -u bench.py --workload config4 --docs 10000000 --terms 1000000 --triples 32 --steps 8 --warmup 3 --no-cpu --no-extra --no-latency 2>gpurun_out/e.err | line; grep -E "Traceback|Error" gpurun_out/e.err | head
done
