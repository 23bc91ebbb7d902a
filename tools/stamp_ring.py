"""Diagnostic: where the waves of k_scan_ring spend their time on the headline workload (stamp build: make -C veloci_amd/csrc stamp).
VQ_LIB=veloci_amd/libveloci_amd_stamp.so python tools/stamp_ring.py [docs] [triples]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
triples = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spec = synth.SynthSpec(num_docs=docs, num_terms=100_000, triples=triples, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False, background_terms=0)
data, meta = synth.generate(spec, device="cuda:0")
idx = veloci_amd.Index(data)
reqs = [veloci_amd.Request(synth.req_and(list(meta.triples[i % triples]), top=10)) for i in range(1024)]
batch = veloci_amd.RequestBatch(reqs)
L = veloci_amd.lib()
buf = (C.c_ulonglong * 32)()
veloci_amd.search_batch_flat(batch, idx, stride=10)
L.vq_debug_ring_stamps(buf, 1)
t0 = time.perf_counter()
veloci_amd.search_batch_flat(batch, idx, stride=10)
dt = time.perf_counter() - t0
L.vq_debug_ring_stamps(buf, 1)
v = list(buf)
cons = {0: "span start (descriptors, directory, first requests)", 1: "service (own loads waited for, flush stage, threshold word)", 2: "slot wait + reads of the tile's postings",
        3: "probe (cover postings against the slot's words)", 4: "rank + release", 5: "next request", 6: "span end (drain, keys out)"}
tiles = max(v[8], 1)
tot = sum(v[k] for k in cons)
print(f"batch of 1024: {dt * 1e3:.2f} ms; consumer waves by phase (shader cycles), {tiles} tiles, {v[11]} spans:")
for k, n in cons.items():
    print(f"  {n:62s} {v[k] / tot * 100:6.2f}%   {v[k] / tiles:9.0f} cycles/tile")
print(f"  total {tot / tiles:.0f} cycles per tile and consumer; tile not there yet at first look: {v[9] / tiles:.3f} of the tiles; tiles with live hits {v[10] / tiles:.3f}")
lt = max(v[19], 1)
load = {16: "idle / polling", 17: "request read + issue", 18: "wait for the tile kRingM back", 20: "publish while idle (wait for the oldest)"}
ltot = sum(v[k] for k in load)
print(f"loader waves, {lt} tiles:")
for k, n in load.items():
    print(f"  {n:62s} {v[k] / max(ltot, 1) * 100:6.2f}%   {v[k] / lt:9.0f} cycles/tile")
print(f"  total {ltot / lt:.0f} cycles per tile and loader; idle polls {v[21]}")
