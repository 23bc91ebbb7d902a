run() { echo "== $1"; env $1 timeout -k 10 300 python -u bench.py --workload or8 --steps 6 --warmup 2 --no-cpu --no-extra --no-latency 2>&1 | grep "^\[bench\] timed"; }
run "VQ_X=1"
run "VQ_SPAN_POSTINGS=131072"
run "VQ_SPAN_POSTINGS=524288"
run "VQ_SPAN_POSTINGS=1048576"
run "VQ_CAND_CAP=32"
run "VQ_CAND_CAP=128"
