# FETCH_SIZE calibration on known byte counts (tools/fetch_calib.hip) -> gpurun_out/<tag>/fetch_calib.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-calib}
O=gpurun_out/$tag; mkdir -p $O
S=/tmp/fetch_calib; rm -rf $S
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $S -o c -- ./tools/fetch_calib > $O/fetch_calib_run.txt 2> $O/fetch_calib.log
python3 tools/pmc_summary.py $S $S/sum.csv > /dev/null
python3 - $S/sum.csv $O/fetch_calib_run.txt > $O/fetch_calib.txt <<'PY'
import csv, sys
known = {}
for line in open(sys.argv[2]):
    if line.startswith("KNOWN"):
        _, k, v = line.split()
        known[k] = int(v)
print("kernel, launches, FETCH_SIZE KB per launch, known bytes per launch (64 B per gathered sector), FETCH_SIZE*1024 / known")
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0]
    for k, v in known.items():
        if name.strip().endswith(k) or name.strip() == k:
            mean = float(row["Mean_per_dispatch"])
            print(f"{k}, {row['Dispatches']}, {mean:.1f}, {v}, {mean * 1024 / v:.4f}")
PY
cat $O/fetch_calib.txt; grep "rep 2" $O/fetch_calib_run.txt
