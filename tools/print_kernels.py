"""Print the per-kernel table of a bench.py JSON line found in a log: python tools/print_kernels.py <log>"""
import json
import sys

for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(f"{d['value']:.0f} {d['unit']}, {d['ms_per_step']} ms per step")
        for k, v in d.get("kernels", {}).items():
            print(f"  {k:28s} {v['launch_ms']:8.4f} ms x {v['launches']:4d}  q/launch {v['queries_per_launch']:7.1f}  {v['GBps']:7.1f} GB/s")
