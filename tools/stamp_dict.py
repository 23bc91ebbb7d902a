"""Diagnostic: phase shares of k_dict_scan (stamp build: make -C veloci_amd/csrc stamp).  VQ_LIB=veloci_amd/libveloci_amd_stamp.so python tools/stamp_dict.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

spec = synth.SynthSpec(num_docs=200000, num_terms=1_000_000, triples=32, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False, background_terms=2000)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
pool = [t for tri in meta.triples for t in tri] + list(meta.background)
probes = bench.edited_terms(pool, 256)
reqs = [veloci_amd.Request({"search_req": {"search": {"terms": [t], "path": "body", "levenshtein_distance": 2}}, "top": 10}) for t in probes]
L = veloci_amd.lib()
buf = (C.c_ulonglong * 16)()
veloci_amd.search_batch(reqs, idx)
L.vq_debug_stamps(buf, 1)
veloci_amd.search_batch(reqs, idx)
L.vq_debug_stamps(buf, 1)
v = list(buf)
names = {0: "prologue (tables, first prefetch)", 1: "round start (prefetch wait, LDS fill)", 2: "phase A (filters, queue)", 3: "phase B (recurrence, matches)"}
tot = sum(v[k] for k in names)
for k, n in names.items():
    print(f"{n:40s} {v[k] / tot * 100:6.2f}%   {v[k] / max(v[7], 1):10.1f} ticks/round")
print("rounds", v[7], "blocks", v[8], "pairs/round", v[11] / max(v[7], 1), "ticks/round total", tot / max(v[7], 1), "(100 MHz memtime ticks of wave 0 of each block)")
