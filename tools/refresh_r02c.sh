# Config #5's own workload (mixed AND / OR + boosts) on a 1/8 shard through the collective path and on the 100 M-doc index; kernel stats of the
# bench_jmdict shape under rocprofv3.  Outputs: gpurun_out/r02c/ (copy to profiles/).
set -x
O=gpurun_out/r02c; mkdir -p $O
VQ_BENCH_COLLECTIVE=1 timeout -k 10 400 python3 bench.py --workload mix --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra > $O/r02_bench_mix_shard8_collective_path.json 2> $O/mix_shard8.err
timeout -k 10 500 python3 bench.py --workload mix --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_mix_100m.json 2> $O/mix_100m.err
cd /tmp && export TMPDIR=/tmp
JM_CACHE=/tmp/jm.pkl CPU=0 STEPS=10 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jmprof -- python3 $GRAFT_REPO_ROOT/tests/bench_jmdict_shape.py > $GRAFT_REPO_ROOT/$O/jm_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/jm_rocprof.err
cd $GRAFT_REPO_ROOT
f=$(find /tmp/jmprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" > $O/r02_jmdict_kernel_stats.csv
ls -la $O
