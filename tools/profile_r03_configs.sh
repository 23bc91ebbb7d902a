# Round-3 counter evidence for the dominant kernels of configs #2, #3 and #4 (k_scan_union, k_scan_simple<2,rich>, k_dict_scan, k_union): rocprofv3 kernel
# stats, FETCH_SIZE / WRITE_SIZE and the SQ instruction / cycle sets, one pass each.  usage: bash tools/profile_r03_configs.sh [config2|config3|config4 ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_configs; mkdir -p $O
prof() {  # tag, bench args...
    tag=$1; shift
    B="bench.py --steps 6 --warmup 2 --no-cpu --no-extra --no-latency --no-parity $*"
    S=/tmp/r03c_${tag}_stats; rm -rf $S
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --kernel-include-regex "vq::" --output-format csv -d $S -o p -- python3 $B > $O/${tag}_bench.json 2> $O/${tag}_stats.err || return 1
    f=$(find $S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
    for set in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" "sq_cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
        set -- $set; name=$1; shift
        S=/tmp/r03c_${tag}_$name; rm -rf $S
        timeout -k 10 300 rocprofv3 --kernel-trace --kernel-include-regex "vq::" --output-format csv --pmc "$@" -d $S -o p -- python3 $B > /dev/null 2> $O/${tag}_$name.err || return 1
        python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null 2>&1
        [ -f $S/sum.csv ] && { head -1 $S/sum.csv > $O/${tag}_pmc_$name.csv; grep "vq::" $S/sum.csv >> $O/${tag}_pmc_$name.csv; }
    done
}
[ $# -eq 0 ] && set -- config3 config4
for c in "$@"; do
    case $c in
        config2) prof config2 --workload single --docs 100000000 || exit 1;;  # (the default run's config2_100m_docs leg: single-term scans over the 256 triples' lists)
        config3) prof config3 --workload config3 --docs 10000000 --terms 1000000 --triples 32 || exit 1;;
        config4) prof config4 --workload config4 --docs 10000000 --terms 1000000 --triples 32 || exit 1;;
    esac
done
ls $O
