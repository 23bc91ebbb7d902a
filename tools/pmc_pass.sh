# usage: tools/pmc_pass.sh <tag> "<COUNTER1 COUNTER2 ...>" <bench args...>  -> gpurun_out/pmc_<tag>.csv (library kernels, mean per dispatch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; ctrs=$2; shift; shift
S=/tmp/pmc_$tag; rm -rf $S
rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $S -o p -- python3 bench.py "$@" > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.log
python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null
head -1 $S/sum.csv > gpurun_out/pmc_$tag.csv; grep "vq::" $S/sum.csv >> gpurun_out/pmc_$tag.csv
cat gpurun_out/pmc_$tag.csv
