"""Diagnostic: phase shares of k_scan_probe on the headline workload (stamp build: make -C veloci_amd/csrc stamp).
VQ_LIB=veloci_amd/libveloci_amd_stamp.so python tools/stamp_probe.py [docs] [triples]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
triples = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spec = synth.SynthSpec(num_docs=docs, num_terms=100_000, triples=triples, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False, background_terms=0)
data, meta = synth.generate(spec, device="cuda:0")
idx = veloci_amd.Index(data)
reqs = [veloci_amd.Request(synth.req_and(list(meta.triples[i % triples]), top=10)) for i in range(1024)]
batch = veloci_amd.RequestBatch(reqs)
L = veloci_amd.lib()
buf = (C.c_ulonglong * 16)()
veloci_amd.search_batch_flat(batch, idx, stride=10)
L.vq_debug_probe_stamps(buf, 1)
t0 = time.perf_counter()
veloci_amd.search_batch_flat(batch, idx, stride=10)
dt = time.perf_counter() - t0
L.vq_debug_probe_stamps(buf, 1)
v = list(buf)
names = {0: "span start (descriptors, directory, first loads)", 1: "tile top (wait for the loads in flight, LDS fill, next loads out)", 4: "flush service", 5: "threshold adoption",
         2: "probe (cover postings against the LDS tiles)", 3: "rank step", 6: "span end (drain, keys out)"}
tot = sum(v[k] for k in names)
tiles = max(v[8], 1)
print(f"batch of 1024: {dt * 1e3:.2f} ms; waves' time by phase (s_memtime ticks = shader cycles), {tiles} tiles:")
for k, n in names.items():
    print(f"  {n:60s} {v[k] / tot * 100:6.2f}%   {v[k] / tiles:9.0f} cycles/tile")
if v[15]:
    print(f"  shader clock while the spans ran: {v[14] / v[15] * 100:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick)")
print(f"  total {tot / tiles:.0f} cycles per tile and wave; rounds/tile {v[9] / tiles:.2f}, rank steps/tile {v[10] / tiles:.3f}, flush services/tile {v[11] / tiles:.3f}, "
      f"final stages/tile {v[12] / tiles:.3f}, pool merges/tile {v[13] / tiles:.3f}")
