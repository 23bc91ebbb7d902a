# bench_jmdict shape under tile / span tuning knobs (one corpus build per box).  Output: gpurun_out/jm_sweep.txt
export JM_CACHE=/tmp/jm_cache.pkl CPU=0 STEPS=8 KERNELS=1
O=gpurun_out/jm_sweep.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 300 python3 tests/bench_jmdict_shape.py 2> /tmp/jm.err | cut -c1-260 >> $O; grep -o '"k_tile_scan": {"ms": [0-9.]*' /tmp/jm.err >> $O; grep -o '"k_union<[a-z]*>": {"ms": [0-9.]*' /tmp/jm.err >> $O; }
run X=0
run VQ_TILE_LDS_KB=24
run VQ_TILE_LDS_KB=48
run VQ_TILE_LDS_KB=48 VQ_TILE_WORDS_MAX=512
run VQ_SPAN_POSTINGS=65536
run VQ_SPAN_POSTINGS=16384
run VQ_TILE_LDS_KB=48 VQ_SPAN_POSTINGS=32768
cat $O
