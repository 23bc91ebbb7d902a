#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy report from the compiler (no GPU needed).
cd "$(dirname "$0")/../veloci_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -Rpass-analysis=kernel-resource-usage -c kernels.hip -o /tmp/kernel_resources.o 2>&1 |
python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    print("%-70s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-6s occupancy %-3s LDS %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
