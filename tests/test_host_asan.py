"""`make asan` (veloci_amd/csrc/Makefile): the host side of the library — index staging, request parsing, query compilation, the C ABI — built with
g++ -fsanitize=address,undefined against a stubbed device layer (tests/native/hip_stub.cpp) and driven over the request fixtures
(tests/native/asan_driver.py).  GPU sanitizers are not available on the pool; this is the part of the product that can run under one."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_side_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "veloci_amd", "csrc"), "-j6", "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    so = os.path.join(ROOT, "veloci_amd", "_host_asan", "libveloci_host_asan.so")
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libstdcxx = subprocess.run(["g++", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(so) and os.path.sep in libasan
    # (libstdc++ preloaded as well: the interpreter does not link it, and ASan resolves __cxa_throw when it starts)
    env = dict(os.environ, VQ_LIB=so, LD_PRELOAD=libasan + " " + libstdcxx, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               VQ_HOST_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(HERE, "native", "asan_driver.py")], capture_output=True, text=True, timeout=900, env=env)
    tail = r.stdout[-1500:] + r.stderr[-6000:]
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    stats = json.loads(r.stdout.split("ASAN_DRIVER_OK ", 1)[1])
    # whole pipelined steps over a "device" whose launches do nothing (compile, pack, launch calls, result assembly, copy-out)
    r2 = subprocess.run([sys.executable, os.path.join(HERE, "native", "asan_step_driver.py")], capture_output=True, text=True, timeout=900, env=dict(env, VQ_STUB_NOOP_LAUNCH="1"))
    tail2 = r2.stdout[-1500:] + r2.stderr[-6000:]
    assert r2.returncode == 0 and "ASAN_STEP_DRIVER_OK" in r2.stdout, tail2
    assert "ERROR: AddressSanitizer" not in r2.stderr and "runtime error:" not in r2.stderr, tail2
    assert json.loads(r2.stdout.split("ASAN_STEP_DRIVER_OK ", 1)[1])["steps"] >= 4
    assert stats["parsed"] > 100 and stats["compiled"] > 70 and stats["ready"] > 30 and stats["device_errors"] > 10 and stats["highlighted"] > 80 and stats["highlight_errors"] > 5, stats
