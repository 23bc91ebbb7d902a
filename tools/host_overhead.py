"""Diagnostic: where does the host time of one batch go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import veloci_amd
from veloci_amd import synth
spec = synth.SynthSpec(num_docs=100_000_000, num_terms=10000, triples=8, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
reqs = [veloci_amd.Request(synth.req_and(list(meta.triples[i % 8]))) for i in range(1024)]
batch = veloci_amd.RequestBatch(reqs)
for k in range(6):
    t0 = time.perf_counter()
    out = veloci_amd.search_batch_flat(batch, idx, stride=10)
    t1 = time.perf_counter()
    print(f"flat call {1e3*(t1-t0):.3f} ms", file=sys.stderr)
for k in range(3):
    t0 = time.perf_counter()
    out = veloci_amd.search_batch(reqs, idx)
    t1 = time.perf_counter()
    print(f"object call {1e3*(t1-t0):.3f} ms", file=sys.stderr)
