# one call: where the secondary workloads stand.  usage: bash tools/state_r02.sh
B="timeout -k 10 400 python -u bench.py --steps 6 --warmup 2 --no-cpu --no-extra --no-latency"
for w in or8 and_of_or4; do
  echo "== $w"; $B --workload $w 2>&1 | grep -E "^\[bench\] timed region|Traceback|Error"
done
echo "== config4 10M"; $B --workload config4 --docs 10000000 --terms 1000000 2>&1 | grep -E "^\[bench\] timed region|Traceback|Error"
echo "== jmdict"; timeout -k 10 400 python -u tests/bench_jmdict_shape.py 2>&1 | tail -6
echo "== full vocabulary"; timeout -k 10 900 python -u tools/full_vocab_footprint.py 2>&1 | tail -3
