# Re-take the plain bench lines of round 2 after the last kernel / host changes (the rocprofv3 passes of tools/profile_r02.sh cover the
# headline kernel, which did not change).  Outputs: gpurun_out/r02b/*.json (copy to profiles/).
set -x
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/r02_bench_n1.json 2> $O/bench_n1.err
for w in or8 and_of_or4; do timeout -k 10 400 python3 bench.py --workload $w --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_${w}_100m.json 2> $O/bench_$w.err; done
timeout -k 10 400 python3 bench.py --workload config4 --docs 10000000 --terms 1000000 --triples 32 --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_config4_10m_32triples.json 2> $O/bench_config4.err
timeout -k 10 400 python3 bench.py --workload config4 --docs 10000000 --terms 1000000 --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_config4_10m.json 2> $O/bench_config4b.err
VQ_BENCH_COLLECTIVE=1 timeout -k 10 400 python3 bench.py --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra > $O/r02_bench_shard8_collective_path.json 2> $O/bench_shard8.err
timeout -k 10 600 python3 tests/bench_jmdict_shape.py > $O/r02_bench_jmdict_shape.json 2> $O/bench_jmdict.err
ls -la $O
