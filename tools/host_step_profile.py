"""Host cost of one step (vq_shard_step_begin / _end) on THIS machine (no GPU): the product's host side built against the stubbed device layer
(tests/native/hip_stub.cpp) with VQ_STUB_NOOP_LAUNCH=1 — launches do nothing, "results" are garbage — runs compile, pack, the launch calls and the
result assembly of 1024-request steps; VQ_TIMING prints where the time went.  A development aid; nothing here is a measurement of the product.
    g++ -std=c++17 -O2 -fPIC -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -o /tmp/libveloci_host.so \
        veloci_amd/csrc/{index,compile,exec,hostpool,capi}.cpp tests/native/hip_stub.cpp
    python tools/host_step_profile.py /tmp/libveloci_host.so [single|and|mix] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VQ_LIB"] = sys.argv[1]
os.environ["VQ_STUB_NOOP_LAUNCH"] = "1"
os.environ.setdefault("VQ_TIMING", "1")
import veloci_amd  # noqa: E402
from veloci_amd import synth  # noqa: E402

shape = sys.argv[2] if len(sys.argv) > 2 else "single"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
spec = synth.SynthSpec(num_docs=1_000_000, num_terms=100_000, triples=8, extra_probe_dfs=(1000, 100_000, 1_000_000), background_terms=0,
                       with_t2t=shape == "mix", with_facets=False, with_boost=shape == "mix", with_phrase=shape == "mix")
data, meta = synth.generate(spec, device="cpu")
idx = veloci_amd.Index(data, device=0)
if shape == "single":
    reqs = [synth.req_single(meta.extra_probes[i % 3]) for i in range(1024)]
elif shape == "and":
    reqs = [synth.req_and(list(meta.triples[i % 8])) for i in range(1024)]
else:
    reqs = [(synth.req_and, synth.req_or, synth.req_and_phrase_locality)[i % 3](list(meta.triples[i % 8])) for i in range(1024)]
batch = veloci_amd.RequestBatch([veloci_amd.Request(r) for r in reqs])
from veloci_amd import dist  # noqa: E402

prev = None
t0 = time.perf_counter()
for s in range(steps):
    h = dist.shard_step_begin(idx, batch)
    if prev is not None:
        dist.shard_step_end(prev, 10)
    prev = h
dist.shard_step_end(prev, 10)
print(f"{steps} steps of 1024 '{shape}' requests: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step (host only, launches stubbed)", file=sys.stderr)
