"""Host cost of a batch of the reference's bench_jmdict requests on THIS machine (no GPU): the product's host side built without the device
(`tests/native/hip_stub.cpp`, prefix probes answered by the stub's loop) runs dictionary scans + compilation pass 1 and stops at the first kernel
launch; VQ_TIMING / VQ_TIMING_SUB print where the time went.  A development aid for the request compiler; nothing here is a measurement of the product.
    python tools/host_profile.py <host-only .so> [batches]"""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["VQ_LIB"] = sys.argv[1]
os.environ.setdefault("VQ_STUB_DICT_SCAN", "1")
os.environ.setdefault("VQ_TIMING", "1")
os.environ.setdefault("VQ_TIMING_SUB", "1")
import veloci_amd  # noqa: E402
from veloci_amd import mini_indexer  # noqa: E402
import bench_jmdict_shape as J  # noqa: E402

cache = "/tmp/jmdict_host_profile.pkl"
if os.path.exists(cache):
    with open(cache, "rb") as f:
        data, terms = pickle.load(f)
else:
    docs, terms = J.corpus(int(os.environ.get("DOCS", "166600")))
    data, info = mini_indexer.build_index(docs, J.INDICES)
    with open(cache, "wb") as f:
        pickle.dump((data, terms), f, protocol=4)
idx = veloci_amd.Index(data, device=0)
reqs = [veloci_amd.Request(J.jmdict_request(terms[i % len(terms)], 0)) for i in range(256)]
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    t0 = time.perf_counter()
    try:
        veloci_amd.search_batch(reqs, idx)
    except veloci_amd.VelociError as e:
        print("stopped at:", str(e)[:80], f"after {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
