"""Diagnostic: phase shares of k_scan_simple / k_scan_simple<2,rich> (stamp build: make -C veloci_amd/csrc stamp; the counters are k_tile_scan's, which
these workloads do not run).  VQ_LIB=veloci_amd/libveloci_amd_stamp.so python tools/stamp_simple.py <and|or|config3|and_of_ors> [docs] [triples]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "and"
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_000
tri = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rich = kind in ("config3", "and_of_ors")
spec = synth.SynthSpec(num_docs=docs, num_terms=100_000, triples=tri, with_t2t=rich, with_facets=False, with_boost=rich, with_phrase=rich, background_terms=0)
data, meta = synth.generate(spec, device="cuda:0")
idx = veloci_amd.Index(data)
T = lambda i: list(meta.triples[i % tri])
mk = {"and": lambda i: synth.req_and(T(i)), "or": lambda i: synth.req_or(T(i)), "config3": lambda i: synth.req_and_phrase_locality(T(i)),
      "and_of_ors": lambda i: synth.req_and_of_ors(T(i)[:2], [T(i)[2], T(i + 1)[2]])}[kind]
batch = veloci_amd.RequestBatch([veloci_amd.Request(mk(i)) for i in range(1024)])
L = veloci_amd.lib()
buf = (C.c_ulonglong * 16)()
veloci_amd.search_batch_flat(batch, idx, stride=10)
L.vq_debug_stamps(buf, 1)
veloci_amd.search_batch_flat(batch, idx, stride=10)
L.vq_debug_stamps(buf, 1)
v = list(buf)
names = {0: "span start", 1: "tile head: next tile, first loads out", 2: "waiting for the loads, scattering the id lists into LDS", 3: "presence, popcounts, mask pruning",
         4: "ranks (popcounts + scans per list)", 5: "survivor loop (queueing)", 10: "flushes (gathers, scoring, candidate buffer)", 6: "span end"}
tot = sum(v[k] for k in names)
tiles = max(v[7], 1)
print(f"{kind} on {docs} docs: {tiles} tiles, {v[8]} spans; waves' time by phase (shader cycles per tile):")
for k, n in names.items():
    print(f"  {n:58s} {v[k] / tot * 100:6.2f}%   {v[k] / tiles:9.0f}")
print(f"  total {tot / tiles:.0f} cycles per tile and wave; flushes per tile {v[9] / tiles:.3f}")
