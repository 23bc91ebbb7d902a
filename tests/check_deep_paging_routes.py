"""One-off check: deep paging through every kernel route (plain / rich / leaf-f32 / generic) against the oracle."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, veloci_amd
from veloci_amd import synth
from oracle import binding as O
from parity import assert_same
spec = synth.SynthSpec(num_docs=300_000, num_terms=5000, triples=2, extra_probe_dfs=(1000, 30_000, 300_000), background_terms=40)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data, device=0)
ora = O.OracleIndex(data.num_anchors); data.load_into(ora)
a = list(meta.triples[0]); p = meta.extra_probes
shapes = [synth.req_single(p[0]), {"search_req": {"search": {"path": "body", "terms": [p[0][:3]], "starts_with": True}}},
          {"search_req": {"search": {"path": "body", "terms": [p[1]], "levenshtein_distance": 1}}}, synth.req_and(a), synth.req_or([p[0], a[2]]),
          dict(synth.req_single(p[0]), boost=[{"path": "pop", "boost_fun": "Add", "param": 1.0}]),
          {"search_req": {"or": {"queries": [{"search": {"path": "body", "terms": [t]}} for t in [p[0]] + list(meta.background[:6])]}}}]
n = 0
for sh in shapes:
    for top, skip in ((3000, 0), (10, 1020), (1500, 900), (5, 100000)):
        r = dict(sh, top=top, skip=skip)
        assert_same(r, veloci_amd.search(r, idx), ora.search_json(json.dumps(r)))
        n += 1
print("ok", n, os.environ.get("VQ_FORCE_GENERIC"), os.environ.get("VQ_NO_LEAF_F32"))
