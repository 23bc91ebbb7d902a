"""Diagnostic: phase shares of k_tile_scan (stamp build).  VQ_LIB=veloci_amd/libveloci_amd_stamp.so python tools/stamp_report.py"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import veloci_amd
from veloci_amd import synth
docs = int(os.environ.get("DOCS", "100000000")); tri = int(os.environ.get("TRIPLES", "8"))
kind = os.environ.get("KIND", "and")  # and | config3 | and_of_ors
rich = kind in ("config3", "and_of_ors")
spec = synth.SynthSpec(num_docs=docs, num_terms=10000, triples=tri, with_t2t=rich, with_facets=False, with_boost=rich, with_phrase=rich)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
T = lambda i: list(meta.triples[i % tri])
mk = {"and": lambda i: synth.req_and(T(i)), "config3": lambda i: synth.req_and_phrase_locality(T(i)),
      "and_of_ors": lambda i: synth.req_and_of_ors(T(i)[:2], [T(i)[2], T(i + 1)[2]]),
      "or8": lambda i: synth.req_or((T(i) + T(i + 1) + T(i + 2))[:8]),
      "and_of_or4": lambda i: {"search_req": {"and": {"queries": [synth.req_or((T(i) + T(i + 1))[:4])["search_req"], synth.req_or((T(i + 1) + T(i + 2))[1:5])["search_req"]]}}, "top": 10}}[kind]
reqs = [veloci_amd.Request(mk(i)) for i in range(int(os.environ.get("BATCH", "256")))]
L = veloci_amd.lib()
buf = (C.c_ulonglong * 16)()
veloci_amd.search_batch(reqs, idx)
L.vq_debug_stamps(buf, 1)
veloci_amd.search_batch(reqs, idx)
L.vq_debug_stamps(buf, 1)
v = list(buf)
names = {0: "init", 1: "P0 next tile + P1 clear", 9: "P2 issue first loads", 10: "P2 scatter / copy", 2: "P2 barrier", 3: "P3 presence", 4: "P4 prefix", 5: "P5 eval", 6: "final"}
tot = sum(v[k] for k in names)
for k, n in names.items():
    print(f"{n:26s} {v[k]/tot*100:6.2f}%   {v[k]/max(v[7],1):10.1f} ticks/tile")
print("tiles", v[7], "wgs", v[8], "ticks/tile total", tot / max(v[7], 1))
print("hits/tile", v[11] / max(v[7], 1), "scored/tile", v[12] / max(v[7], 1))
print("P5 scoring rounds of lane 0 per tile", v[13] / max(v[7], 1), "ticks per round in the score tree", v[14] / max(v[13], 1), "prunes in P5 per tile", v[15] / max(v[7], 1))
