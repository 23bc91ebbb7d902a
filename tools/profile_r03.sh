# Counter evidence for the headline kernel (round 3): rocprofv3 kernel stats + SQ / TCC / TCP passes of the bench command, each in its own
# pass (never --pmc together with a trace domain other than --kernel-trace).  Raw output under /tmp on the GPU box, per-kernel means under
# gpurun_out/r03_<tag>/ (copy what is to be judged into profiles/).
# usage: tools/profile_r03.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/r03_$tag; mkdir -p $O
B="bench.py --steps 4 --warmup 2 --no-cpu --no-extra --no-latency $*"
echo "== $tag: $B" > $O/log.txt
[ -f gpurun_out/r03_counters.txt ] || rocprofv3 -L > gpurun_out/r03_counters.txt 2>&1
run() {  # name, counters...
    name=$1; shift
    S=/tmp/r03_${tag}_$name; rm -rf $S
    timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc "$@" -d $S -o p -- python3 $B > $O/bench_$name.json 2> $O/bench_$name.err
    rc=$?
    echo "pass $name rc=$rc" >> $O/log.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping" >> $O/log.txt; exit 1; fi
    python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1
    if [ -f $S/sum.csv ]; then head -1 $S/sum.csv > $O/pmc_$name.csv; grep "vq::" $S/sum.csv >> $O/pmc_$name.csv; fi
    tail -2 $O/bench_$name.err >> $O/log.txt
}
S=/tmp/r03_${tag}_stats; rm -rf $S
timeout -k 10 300 rocprofv3 --kernel-trace --stats --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 $B > $O/bench_stats.json 2> $O/bench_stats.err
rc=$?; echo "stats rc=$rc" >> $O/log.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
f=$(find $S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
run sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES
run sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcc_req TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
cat $O/log.txt
for f in $O/pmc_*.csv; do echo "-- $f"; grep "k_scan\|k_and" $f; done
exit 0
