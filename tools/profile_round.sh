# One round of measurements on the GPU box: bench lines, rocprofv3 kernel stats, HBM traffic counters (separate passes).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
S=/tmp/prof_scratch
rm -rf $S && mkdir -p $O $S
python3 bench.py --cpu-seconds 5 > $O/bench_and2.json 2> $O/bench_and2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats -o stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof_stats.log
find $S/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $S/pmc_fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.log
python3 tools/pmc_summary.py $S/pmc_fetch $O/pmc_fetch_size.csv
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $S/pmc_write -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > $O/bench_pmc_write.json 2> $O/pmc_write.log
python3 tools/pmc_summary.py $S/pmc_write $O/pmc_write_size.csv
du -sh $O
ls $O
