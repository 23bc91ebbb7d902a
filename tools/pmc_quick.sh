cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/r04_$tag; mkdir -p $O
B="bench.py --steps 4 --warmup 2 --no-cpu --no-extra --no-latency $*"
run() { name=$1; shift; S=/tmp/r04_${tag}_$name; rm -rf $S
  timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc "$@" -d $S -o p -- python3 $B > $O/bench_$name.json 2> $O/bench_$name.err || exit 1
  python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1; head -1 $S/sum.csv > $O/pmc_$name.csv; grep "vq::" $S/sum.csv >> $O/pmc_$name.csv; grep "k_scan" $O/pmc_$name.csv; }
run sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
run sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run sq_more SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH
