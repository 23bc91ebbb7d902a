# host compile time of a 1024-query batch vs host threads and worker polling (VQ_TIMING pass 1).  usage: bash tools/compile_scaling.sh
for spin in 0 3000; do
for t in 4 16 32; do
  echo "== threads=$t spin_us=$spin"
  env VQ_POOL_SPIN_US=$spin VQ_HOST_THREADS=$t VQ_BENCH_COLLECTIVE=1 VQ_SHARD_CHUNKS=1 VQ_TIMING=1 timeout -k 10 200 python -u bench.py --docs 12500000 --steps 8 --warmup 3 --no-cpu --no-extra --no-latency > /tmp/cs.log 2>&1
  grep -E "vq timing\] n=" /tmp/cs.log | tail -3 | cut -c1-48
  grep -E "^\[bench\] timed region" /tmp/cs.log
done
done
