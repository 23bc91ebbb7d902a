"""Driver of the CPU sanitizer build (run by tests/test_host_asan.py under LD_PRELOAD=libasan, VQ_LIB=<libveloci_host_asan.so>): stages the
fixture corpora through the C ABI (HIP runtime stubbed to host memory), parses every fixture request text, compiles every fixture request
(vq_debug_compile: dictionary lookups, list tables, op programs, layouts — everything in front of the first kernel launch) and walks the error
paths of a search whose first launch throws.  Prints one summary line; any sanitizer report aborts the process."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)

import refcases  # noqa: E402
import veloci_amd  # noqa: E402
from veloci_amd import _lib, mini_indexer  # noqa: E402

L = _lib.lib()
assert "host_asan" in _lib.lib_path(), _lib.lib_path()
stats = {"parsed": 0, "parse_errors": 0, "compiled": 0, "ready": 0, "prepass": 0, "declined": 0, "device_errors": 0}

with open(os.path.join(TESTS, "golden", "request_parse.json"), encoding="utf-8") as f:
    for c in json.load(f)["cases"]:
        h = C.c_void_p()
        raw = c["text"].encode("utf-8")
        rc = L.vq_request_parse(raw, len(raw), C.byref(h))
        if rc == 0:
            assert L.vq_request_to_json(h).decode("utf-8") == c["parsed"]
            L.vq_request_free(h)
            stats["parsed"] += 1
        else:
            assert "error" in c
            stats["parse_errors"] += 1


def compile_all(index, requests):
    for r in requests:
        req = veloci_amd.Request(r)
        st = L.vq_debug_compile(index.h, req.h)
        stats["compiled"] += 1
        stats["ready" if st == 0 else "prepass" if st < 0 else "declined"] += 1


fx = refcases.load()
by_corpus = {}
for c in fx["cases"]:
    if "request" in c:
        by_corpus.setdefault(c["corpus"], []).append(c["request"])
with open(os.path.join(TESTS, "golden", "reference_explain.json"), encoding="utf-8") as f:
    by_corpus.setdefault("test_all", []).extend(c["request"] for c in json.load(f)["cases"])
for name, reqs in by_corpus.items():
    data, docs, info = refcases.build(name)
    idx = veloci_amd.Index(data, device=0)
    compile_all(idx, reqs)
    # a search gets as far as its first launch (or pre-pass) and must unwind cleanly: workspaces released, no leak of the batch
    for r in reqs[:6]:
        try:
            veloci_amd.search(r, idx)
        except veloci_amd.VelociError as e:
            stats["device_errors"] += e.kind == "Device"
    try:  # a device failure fails the whole batch
        veloci_amd.search_batch(reqs[:8], idx, raise_on_error=False)
    except veloci_amd.VelociError as e:
        stats["device_errors"] += e.kind == "Device"
    del idx

# highlight (search_field.rs:233-245) with exact and prefix terms is host work from end to end here (the stub answers prefix probes with a loop):
# snippets under the sanitizers, compared with the oracle's
import test_reference_integration as T  # noqa: E402
from oracle import binding as O  # noqa: E402
data, docs, info = refcases.build("test_all")
idx = veloci_amd.Index(data, device=0)
ora = O.OracleIndex(data.num_anchors)
data.load_into(ora)
stats["highlighted"] = stats["highlight_errors"] = 0
os.environ["VQ_STUB_DICT_SCAN"] = "1"  # prefix probes answered by the stub's plain loop: the host side of a dictionary scan (probe tables, bucketing, scoring) runs too
for part in T.highlight_parts():
    if part.get("levenshtein_distance"):
        continue
    js = json.dumps(part)
    try:
        want = ora.highlight_json(js)
    except O.OracleError as e:
        try:
            veloci_amd.highlight(part, idx)
            raise AssertionError("no error for " + js)
        except veloci_amd.VelociError as g:
            assert str(g) == str(e), (js, str(g), str(e))
        stats["highlight_errors"] += 1
        continue
    assert veloci_amd.highlight(part, idx) == want, js
    stats["highlighted"] += 1
del idx

with open(os.path.join(TESTS, "golden", "reference_query_generator.json"), encoding="utf-8") as f:
    qg = json.load(f)
for name, c in qg["corpora"].items():
    data, info = mini_indexer.build_index(c["docs"], c["indices"], token_values=tuple(c["token_values"]) if c.get("token_values") else None)
    idx = veloci_amd.Index(data, device=0)
    compile_all(idx, [k["request"] for k in qg["cases"] if k["corpus"] == name and "request" in k])
    del idx
print("ASAN_DRIVER_OK " + json.dumps(stats))
