# Round-2 evidence in one call: raw outputs under /tmp on the GPU box, summarised there by tools/summarise_r02.py into gpurun_out/r02/ (copy to profiles/).
# usage (GPU box): bash tools/profile_r02.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=/tmp/r02_raw; rm -rf $O; mkdir -p $O gpurun_out/r02
B="python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --no-latency"
# 1. kernel trace + stats of the headline workload
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# 2. HBM traffic counters, one pass each
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/bench_fetch.json 2> $O/bench_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/bench_write.json 2> $O/bench_write.err
# 3. SQ counters of the 8-leaf OR: k_tile_scan (VQ_NO_WIDE=1) vs k_scan_wide
W="python3 bench.py --workload or8 --steps 2 --warmup 1 --no-cpu --no-extra --no-latency"
VQ_NO_WIDE=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/sq_tile -- $W > $O/or8_tile.json 2> $O/or8_tile.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/sq_wide -- $W > $O/or8_wide.json 2> $O/or8_wide.err
# 4. plain runs
timeout -k 10 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for w in or8 and_of_or4; do timeout -k 10 400 python3 bench.py --workload $w --steps 8 --warmup 3 --no-cpu --no-extra > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout -k 10 400 python3 bench.py --workload config4 --docs 10000000 --terms 1000000 --steps 8 --warmup 3 --no-cpu --no-extra > $O/bench_config4_10m_256triples.json 2> $O/bench_config4.err
VQ_BENCH_COLLECTIVE=1 timeout -k 10 400 python3 bench.py --docs 12500000 --steps 20 --warmup 5 --no-cpu --no-extra > $O/bench_shard8_collective.json 2> $O/bench_shard8.err
timeout -k 10 600 python3 tests/bench_jmdict_shape.py > $O/bench_jmdict_shape.json 2> $O/bench_jmdict.err
timeout -k 10 900 python3 tools/full_vocab_footprint.py > $O/full_vocab_footprint.json 2> $O/full_vocab.err
python3 tools/summarise_r02.py $O gpurun_out/r02
cp $O/*.err gpurun_out/r02/ 2>/dev/null
du -sh gpurun_out/r02
ls gpurun_out/r02
