# The headline workload under rocprofv3 with the round's final code: kernel trace + stats, FETCH_SIZE and WRITE_SIZE in separate passes.
# Raw outputs under /tmp on the GPU box, summarised there by tools/summarise_r02.py into gpurun_out/r02e/ (copy to profiles/).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=/tmp/r02_raw; rm -rf $O; mkdir -p $O gpurun_out/r02e
B="python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --no-latency"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/bench_fetch.json 2> $O/bench_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/bench_write.json 2> $O/bench_write.err
python3 tools/summarise_r02.py $O gpurun_out/r02e
du -sh gpurun_out/r02e
ls gpurun_out/r02e
