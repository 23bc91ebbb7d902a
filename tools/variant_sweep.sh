# occupancy variants of k_scan_simple (make -C veloci_amd/csrc variant NAME=... DEFS=...): config #3 / plain OR / the 1/8-shard AND with each library.
# usage: bash tools/variant_sweep.sh  (on the GPU box)
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['launch_ms'], r['frac'], d.get('parity_checked'))"
}
run() {  # lib, label, bench args...
  lib=$1; shift; echo "== $lib: $*"
  env VQ_LIB=veloci_amd/$lib VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 python -u bench.py --steps 12 --warmup 4 --no-cpu --no-extra --no-latency "$@" 2>gpurun_out/vs.err | line
  grep -E "Traceback|Error" gpurun_out/vs.err | head -3
}
for lib in libveloci_amd.so libveloci_amd_rich3.so; do
  run $lib --workload config3 --docs 10000000 --terms 1000000 --triples 1024
done
for lib in libveloci_amd.so libveloci_amd_simple4.so; do
  run $lib --workload or --docs 100000000 --triples 256
  VQ_BENCH_COLLECTIVE=1 run $lib --workload and --docs 12500000 --triples 256
done
