import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, veloci_amd
from veloci_amd import synth
from oracle import binding as O
spec = synth.SynthSpec(num_docs=300_000, num_terms=5000, triples=2, extra_probe_dfs=(1000, 30_000, 300_000), background_terms=40)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data, device=0)
ora = O.OracleIndex(data.num_anchors); data.load_into(ora)
req = {"search_req": {"search": {"path": "body", "terms": ["nbdcq"]}}, "top": 1200, "skip": 1100, "boost": [{"path": "pop", "boost_fun": "Multiply", "param": 1.0}], "text_locality": True, "boost_term": [{"path": "body", "terms": ["tuzaqw"], "boost": 3.0}]}
for variant in (req, dict(req, top=3000, skip=0), {k: v for k, v in req.items() if k != "boost_term"}, {k: v for k, v in req.items() if k != "boost"}):
    w = ora.search_json(json.dumps(variant)); g = veloci_amd.search(variant, idx)
    gi, wi = g.ids.tolist(), w.ids.tolist()
    first = next((i for i, (a, b) in enumerate(zip(gi, wi)) if a != b), None)
    print("num_hits", g.num_hits, w.num_hits, "len", len(gi), len(wi), "first diff", first, "keys", sorted(variant.keys()))
    if first is not None:
        print(" got ", gi[first-2:first+4], g.scores.tolist()[first-2:first+4]); print(" want", wi[first-2:first+4], w.scores.tolist()[first-2:first+4])
