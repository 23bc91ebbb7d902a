# usage: tools/kstats.sh <tag> <bench args...>   -> gpurun_out/kstats_<tag>.csv (library kernels only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
S=/tmp/kstats_$tag; rm -rf $S
rocprofv3 --kernel-trace --stats --output-format csv -d $S -o s -- python3 bench.py "$@" > gpurun_out/kstats_$tag.json 2> gpurun_out/kstats_$tag.log
f=$(find $S -name "*kernel_stats.csv" | head -1)
head -1 $f > gpurun_out/kstats_$tag.csv; grep "vq::" $f >> gpurun_out/kstats_$tag.csv
cut -d, -f1-4 gpurun_out/kstats_$tag.csv | sed 's/(.*)//' | cut -c1-150
