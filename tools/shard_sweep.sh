# step time of one doc-range shard through the collective path (1 rank, RCCL): chunking / exchange variants.  usage: bash tools/shard_sweep.sh
run() {
  echo "== docs=$1 env=$2"
  env VQ_BENCH_COLLECTIVE=1 $2 timeout -k 10 300 python -u bench.py --docs $1 --steps 20 --warmup 5 --no-cpu --no-extra --no-latency 2>&1 | grep -E "^\[bench\] timed region|Traceback|Error" || true
}
for docs in 12500000; do
  run $docs "VQ_SHARD_CHUNKS=1"
  run $docs "VQ_SHARD_CHUNKS=2"
  run $docs "VQ_SHARD_CHUNKS=2 VQ_PER_CHUNK_COLLECTIVE=1"
  run $docs "VQ_SHARD_CHUNKS=1 VQ_SIMPLE_NV=1"
  run $docs "VQ_SHARD_CHUNKS=2 VQ_SIMPLE_NV=1"
  run $docs "VQ_SHARD_CHUNKS=1 VQ_HOST_THREADS=32"
  run $docs "VQ_SHARD_CHUNKS=1 VQ_HOST_THREADS=8"
done
