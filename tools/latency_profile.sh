# usage: tools/latency_profile.sh <kinds>  -> kernel durations of single-request searches + host timing lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
S=/tmp/lat_prof; rm -rf $S
KINDS=$1 REPS=100 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o s -- python3 tools/latency.py > gpurun_out/latprof.log 2>&1
f=$(find $S -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'vq::' in r['Name']: print(r['Name'].split('(')[0][:40], 'calls', r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'min_us', float(r['MinNs'])/1e3)
"
KINDS=$1 REPS=5 VQ_TIMING=1 python3 tools/latency.py 2>&1 | tail -8
