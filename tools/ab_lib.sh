# A/B of two builds of the library on the headline, alternating on one box.  usage: bash tools/ab_lib.sh <libA.so> <libB.so> [bench args...]
A=$1; B=$2; shift 2
for i in 1 2 3; do for lib in $A $B; do
  echo -n "$lib: "; VQ_LIB=veloci_amd/$lib timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-extra --no-latency --no-cpu --no-parity "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['launch_ms'], r['frac'])"
done; done
