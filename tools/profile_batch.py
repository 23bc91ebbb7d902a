"""Small driver for rocprofv3 counter runs: a few batches of 3-term AND on the 100M-doc index."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import veloci_amd
from veloci_amd import synth
docs = int(os.environ.get("DOCS", "100000000")); tri = int(os.environ.get("TRIPLES", "8")); nb = int(os.environ.get("BATCHES", "2"))
kind = os.environ.get("KIND", "and")
spec = synth.SynthSpec(num_docs=docs, num_terms=10000, triples=tri, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
mk = synth.req_and if kind == "and" else synth.req_or
reqs = [veloci_amd.Request(mk(list(meta.triples[i % tri]))) for i in range(int(os.environ.get("BATCH", "256")))]
for _ in range(nb):
    r = veloci_amd.search_batch(reqs, idx)
print("done", r[0].num_hits)
