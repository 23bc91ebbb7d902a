# Counter evidence for the headline kernel (round 4): rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ and two SQ sets of the
# default bench command, each in its own pass (never --pmc together with a trace domain other than --kernel-trace).  Raw output under /tmp on
# the GPU box, per-kernel means under gpurun_out/r04_<tag>/; tools/traffic_r04.py turns them into profiles/r04_traffic.json.
# usage: tools/profile_r04.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/r04_$tag; mkdir -p $O
B="bench.py --steps 4 --warmup 2 --no-cpu --no-extra --no-latency $*"
echo "== $tag: $B" > $O/log.txt
run() {  # name, counters...
    name=$1; shift
    S=/tmp/r04_${tag}_$name; rm -rf $S
    timeout -k 10 400 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc "$@" -d $S -o p -- python3 $B > $O/bench_$name.json 2> $O/bench_$name.err
    rc=$?
    echo "pass $name rc=$rc" >> $O/log.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping" >> $O/log.txt; exit 1; fi
    python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1
    if [ -f $S/sum.csv ]; then head -1 $S/sum.csv > $O/pmc_$name.csv; grep "vq::" $S/sum.csv >> $O/pmc_$name.csv; fi
    tail -2 $O/bench_$name.err >> $O/log.txt
}
S=/tmp/r04_${tag}_stats; rm -rf $S
timeout -k 10 400 rocprofv3 --kernel-trace --stats --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 $B > $O/bench_stats.json 2> $O/bench_stats.err
rc=$?; echo "stats rc=$rc" >> $O/log.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
f=$(find $S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES
run sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
python3 tools/traffic_r04.py $O > $O/traffic.json 2>> $O/log.txt
cat $O/log.txt
exit 0
