"""Host cost of compiling one request (vq_debug_compile), single thread, on the CPU: python tools/compile_bench.py [path to a host-only build].
The host-only build is the sanitizer recipe of veloci_amd/csrc/Makefile without the sanitizers (device layer stubbed, tests/native/hip_stub.cpp)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    os.environ["VQ_LIB"] = sys.argv[1]
import refcases  # noqa: E402
import veloci_amd  # noqa: E402
from veloci_amd import _lib  # noqa: E402

L = _lib.lib()
data, docs, info = refcases.build("test_all")
idx = veloci_amd.Index(data, device=0)
leaf = lambda t, p: {"search": {"terms": [t], "path": p}}
shapes = {
    "and3": {"search_req": {"and": {"queries": [leaf("will", "meanings.eng[]"), leaf("urge", "meanings.eng[]"), leaf("majestät", "meanings.ger[]")]}}, "top": 10},
    "single": {"search_req": leaf("will", "meanings.eng[]"), "top": 10},
    "or2": {"search_req": {"or": {"queries": [leaf("will", "meanings.eng[]"), leaf("majestät", "meanings.ger[]")]}}, "top": 10},
}
n = int(os.environ.get("N", "100000"))
for name, r in shapes.items():
    req = veloci_amd.Request(r)
    st = L.vq_debug_compile(idx.h, req.h)
    fn, ih, rh = L.vq_debug_compile, idx.h, req.h
    t0 = time.perf_counter()
    for _ in range(n):
        fn(ih, rh)
    dt = time.perf_counter() - t0
    print(f"{name}: status {st}, {dt / n * 1e6:.2f} us per compile (includes ~0.3 us of ctypes call overhead)")
