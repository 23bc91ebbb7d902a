# One round of measurements on the GPU box: bench lines, rocprofv3 kernel stats, HBM traffic counters (separate passes).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01i
S=/tmp/prof_scratch
rm -rf $S && mkdir -p $O $S
python3 bench.py > $O/bench_and.json 2> $O/bench_and.log
python3 bench.py --workload or --no-cpu --steps 5 > $O/bench_or.json 2> $O/bench_or.log
python3 bench.py --workload single --no-cpu --steps 5 > $O/bench_single.json 2> $O/bench_single.log
python3 bench.py --workload config3 --docs 10000000 --no-cpu --steps 5 > $O/bench_config3_10m.json 2> $O/bench_config3.log
python3 bench.py --workload config3 --docs 100000000 --no-cpu --steps 3 > $O/bench_config3_100m.json 2>> $O/bench_config3.log
python3 bench.py --workload and_of_ors --docs 100000000 --no-cpu --steps 3 > $O/bench_and_of_ors_100m.json 2>> $O/bench_config3.log
python3 bench.py --workload mix --docs 100000000 --no-cpu --steps 5 > $O/bench_mix_100m.json 2> $O/bench_mix.log
python3 bench.py --workload config4 --docs 10000000 --terms 1000000 --no-cpu --steps 3 > $O/bench_config4_10m.json 2> $O/bench_config4.log
python3 bench.py --workload or8 --docs 100000000 --no-cpu --steps 3 --no-latency > $O/bench_or8_100m.json 2> $O/bench_or8.log
python3 bench.py --workload and_of_or4 --docs 100000000 --no-cpu --steps 3 --no-latency > $O/bench_and_of_or4_100m.json 2>> $O/bench_or8.log
python3 tools/latency.py > $O/latency_single_request.txt 2> $O/latency.log
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats -o stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof_stats.log
f=$(find $S/stats -name "*kernel_stats.csv" | head -1); head -1 $f > $O/kernel_stats.csv; grep "vq::" $f >> $O/kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $S/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.log
python3 tools/pmc_summary.py $S/pmc_fetch $S/pmc_fetch_size.csv; head -1 $S/pmc_fetch_size.csv > $O/pmc_fetch_size.csv; grep "vq::" $S/pmc_fetch_size.csv >> $O/pmc_fetch_size.csv
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $S/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $O/bench_pmc_write.json 2> $O/pmc_write.log
python3 tools/pmc_summary.py $S/pmc_write $S/pmc_write_size.csv; head -1 $S/pmc_write_size.csv > $O/pmc_write_size.csv; grep "vq::" $S/pmc_write_size.csv >> $O/pmc_write_size.csv
du -sh $O; ls $O
