# End-of-round evidence: every bench line of round 2 re-taken on one box with the final code.  Outputs: gpurun_out/r02d/ (copy to profiles/).
set -x
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/r02_bench_n1.json 2> $O/bench_n1.err
for w in or8 and_of_or4 mix; do timeout -k 10 400 python3 bench.py --workload $w --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_${w}_100m.json 2> $O/bench_$w.err; done
timeout -k 10 400 python3 bench.py --workload config4 --docs 10000000 --terms 1000000 --triples 32 --steps 8 --warmup 3 --no-cpu --no-extra > $O/r02_bench_config4_10m_32triples.json 2> $O/bench_config4.err
VQ_BENCH_COLLECTIVE=1 timeout -k 10 400 python3 bench.py --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra > $O/r02_bench_shard8_collective_path.json 2> $O/bench_shard8.err
VQ_BENCH_COLLECTIVE=1 timeout -k 10 400 python3 bench.py --workload mix --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra > $O/r02_bench_mix_shard8_collective_path.json 2> $O/mix_shard8.err
JM_CACHE=/tmp/jm.pkl KERNELS=1 STEPS=16 timeout -k 10 600 python3 tests/bench_jmdict_shape.py > $O/r02_bench_jmdict_shape.json 2> $O/bench_jmdict.err
grep '"kernels"' $O/bench_jmdict.err > $O/r02_jmdict_kernels.json
ls -la $O
