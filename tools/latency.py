"""p50 / p95 latency of single requests (batch of one) on the synthetic index: DOCS, TRIPLES, KINDS=and,or,single env."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import veloci_amd
from veloci_amd import synth
docs = int(os.environ.get("DOCS", "100000000")); tri = int(os.environ.get("TRIPLES", "8")); reps = int(os.environ.get("REPS", "300"))
spec = synth.SynthSpec(num_docs=docs, num_terms=10000, triples=tri, with_t2t=False, with_facets=False, with_boost=False, with_phrase=False)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data)
makers = {"and": synth.req_and, "or": synth.req_or, "single": lambda t: synth.req_single(t[0])}
for kind in os.environ.get("KINDS", "and,or,single").split(","):
    reqs = [veloci_amd.Request(makers[kind](list(meta.triples[i % tri]))) for i in range(tri)]
    for i in range(20):
        veloci_amd.search(reqs[i % tri], idx)
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        veloci_amd.search(reqs[i % tri], idx)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(f"{kind}: p50 {np.percentile(ts, 50):.3f} ms  p95 {np.percentile(ts, 95):.3f} ms  min {ts.min():.3f} ms  (VQ_SPAN_TARGET={os.environ.get('VQ_SPAN_TARGET', 'default')})", flush=True)
