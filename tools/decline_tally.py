"""Which requests of the randomized differential test does the MI355X path decline, and why?  (GPU box)"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import veloci_amd
import refcases
from test_gpu_parity import _random_request
data, docs, info = refcases.build("test_all")
idx = veloci_amd.Index(data, device=0)
rng = np.random.default_rng(20241003)
why = collections.Counter()
ex = {}
for i in range(600):
    req = _random_request(rng, info)
    try:
        veloci_amd.search(req, idx)
    except veloci_amd.VelociError as e:
        key = (e.kind, str(e)[:110])
        why[key] += 1
        ex.setdefault(key, json.dumps(req)[:400])
print(f"600 random requests (the generator of test_random_requests_on_reference_corpus_match_the_oracle, seed 20241003): {sum(why.values())} declined")
for k, v in why.most_common():
    print(v, k)
    print("    e.g.", ex[k])
