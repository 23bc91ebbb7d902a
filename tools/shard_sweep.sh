# step time of one doc-range shard through the collective path (1 rank, RCCL) vs chunks per step.  usage: bash tools/shard_sweep.sh
run() {
  echo "== docs=$1 env=$2"
  env VQ_BENCH_COLLECTIVE=1 $2 timeout -k 10 300 python -u bench.py --docs $1 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>&1 | grep -E "^\[bench\] timed region|Traceback|Error" || true
}
for docs in 25000000 50000000; do
  for c in 1 2 4; do run $docs "VQ_SHARD_CHUNKS=$c"; done
done
run 12500000 "VQ_SHARD_CHUNKS=1"
run 12500000 "VQ_SHARD_CHUNKS=2"
