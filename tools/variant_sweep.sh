# variants of k_scan_simple (make -C veloci_amd/csrc variant NAME=... DEFS=...): config #3 / the mix / the 1/8-shard AND with each library.
# usage: bash tools/variant_sweep.sh libA.so libB.so  (on the GPU box)
line() {
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
k=d['kernels']
print(d['value'], d['ms_per_step'], {n:(v['launch_ms'],v['queries_per_launch']) for n,v in k.items() if v['scan']})"
}
run() {  # lib, bench args...
  lib=$1; shift; echo -n "$lib $*: "
  env VQ_LIB=veloci_amd/$lib VQ_BENCH_ONE_LAUNCH=1 timeout -k 10 400 python -u bench.py --steps 12 --warmup 4 --no-cpu --no-extra --no-latency --no-parity "$@" 2>gpurun_out/vs.err | line
  grep -E "Traceback|Error" gpurun_out/vs.err | head -3
}
for lib in "$@"; do
  run $lib --workload config3 --docs 10000000 --terms 1000000 --triples 1024
  VQ_BENCH_COLLECTIVE=1 run $lib --workload and --docs 12500000 --triples 256
  run $lib --workload mix --docs 100000000 --triples 256
done
