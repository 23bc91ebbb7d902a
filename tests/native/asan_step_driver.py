"""Second driver of the CPU sanitizer build (tests/test_host_asan.py, with VQ_STUB_NOOP_LAUNCH=1: the stubbed launchers return instead of throwing):
the host side of whole pipelined steps — vq_shard_step_begin / _end with two steps in flight, with and without facets, through the custom exchange of
two shards in one process — runs to the end over whatever the stubbed "device" memory holds: compile, pack, the launch calls, merge bookkeeping,
result assembly and copy-out.  Results are garbage; any sanitizer report aborts the process."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)

import refcases  # noqa: E402
import veloci_amd  # noqa: E402
from veloci_amd import _lib, dist, synth  # noqa: E402

assert "host_asan" in _lib.lib_path(), _lib.lib_path()
assert os.environ.get("VQ_STUB_NOOP_LAUNCH") == "1"
steps = 0
spec = synth.SynthSpec(num_docs=60_000, num_terms=3000, triples=2, extra_probe_dfs=(500, 20_000), background_terms=20)
data, meta = synth.generate(spec, device="cpu")
a, b = meta.triples
reqs = []
for i in range(300):
    t = (a, b)[i & 1]
    reqs.append((synth.req_and(list(t)), synth.req_or(list(t), top=20), synth.req_single(meta.extra_probes[i % 2]), synth.req_and_phrase_locality(list(t)),
                 dict(synth.req_single(meta.background[i % 20]), facets=[{"field": "cat", "top": 5}]))[i % 5])
batch = veloci_amd.RequestBatch([veloci_amd.Request(r) for r in reqs])
index = veloci_amd.Index(data, device=0)
prev = None
for s in range(4):  # two steps in flight on an unsharded index
    h = dist.shard_step_begin(index, batch)
    if prev is not None:
        out = dist.shard_step_end(prev, 20)
        assert len(out[0]) == len(reqs)
        steps += 1
    prev = h
dist.shard_step_end(prev, 20)
steps += 1
del index

# the reference's own test corpora and fixture requests (boosts, filters, 1:n boosts, facets, fuzzy leaves answered by the stub's loop): one step per corpus;
# requests whose pre-passes cannot run over a stubbed device come back with a status, which is fine
os.environ.setdefault("VQ_STUB_DICT_SCAN", "1")
fx = refcases.load()
by_corpus = {}
for c in fx["cases"]:
    if "request" in c and not c["request"].get("explain") and not c["request"].get("why_found") and not c["request"].get("select"):
        by_corpus.setdefault(c["corpus"], []).append(c["request"])
for name, rs in by_corpus.items():
    data2, docs2, info2 = refcases.build(name)
    index2 = veloci_amd.Index(data2, device=0)
    rs = [r for r in rs if int(r.get("top", 10) or 10) + int(r.get("skip", 0) or 0) <= 64][:48]
    if rs:
        sb = veloci_amd.RequestBatch([veloci_amd.Request(r) for r in rs])
        try:
            dist.shard_step_end(dist.shard_step_begin(index2, sb), 64)
            steps += 1
        except veloci_amd.VelociError as e:  # a pre-pass the stub cannot serve fails the step as a whole
            assert e.kind in ("Device", "Unsupported"), e
    del index2
print("ASAN_STEP_DRIVER_OK " + json.dumps({"steps": steps}))
