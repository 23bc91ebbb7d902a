"""Scan time of the config #4 requests with and without their facets (same index, same fuzzy leaves): what the facet rows cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import veloci_amd
from veloci_amd import synth
spec = synth.SynthSpec(num_docs=10_000_000, num_terms=1_000_000, triples=32, with_t2t=False, with_facets=True, with_boost=False, with_phrase=False, background_terms=2000)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
rng = np.random.default_rng(4)
pool = [t for tri in meta.triples for t in tri] + list(meta.background)
alphabet = "abcdefghijklmnopqrstuvwxyz"
def edit(w):
    w = list(w)
    for _ in range(int(rng.integers(1, 3))):
        op = int(rng.integers(0, 3)); pos = int(rng.integers(0, len(w)))
        if op == 0 and len(w) > 3: del w[pos]
        elif op == 1: w.insert(pos, alphabet[int(rng.integers(0, 26))])
        else: w[pos] = alphabet[int(rng.integers(0, 26))]
    return "".join(w)
qterms = [edit(pool[int(rng.integers(0, len(pool)))]) for _ in range(100)]
for name, facets in (("no facets", None), ("cat", [{"field": "cat"}]), ("tags[]", [{"field": "tags[]"}]), ("cat + tags[]", [{"field": "cat"}, {"field": "tags[]"}])):
    reqs = []
    for i in range(256):
        r = {"search_req": {"search": {"path": "body", "terms": [qterms[i % 100]], "levenshtein_distance": 2}}, "top": 10}
        if facets: r["facets"] = facets
        reqs.append(veloci_amd.Request(r))
    res = veloci_amd.search_batch(reqs, idx)
    idx.profile_enable(True); idx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(5): veloci_amd.search_batch(reqs, idx)
    dt = (time.perf_counter() - t0) / 5
    ms, launches, _ = idx.profile_read(reset=True)
    hits = sorted(int(r.num_hits) for r in res)
    print(f"{name:14s} batch {dt*1e3:7.2f} ms  scan launches {ms/launches:6.2f} ms   hits: median {hits[128]} max {hits[-1]} sum {sum(hits)}", flush=True)
