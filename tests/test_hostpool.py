"""The fork-join pool that compiles a batch's requests (veloci_amd/csrc/hostpool.cpp: futex wake-up, atomically claimed parts) under ThreadSanitizer
on the CPU: every part of every job exactly once, no access to a job after run() returned.  (GPU sanitizers are not available on the pool: the host
side is what can be checked this way.)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(flags, out):
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", *flags, "-o", out, os.path.join(HERE, "native", "hostpool_stress.cpp"),
           os.path.join(ROOT, "veloci_amd", "csrc", "hostpool.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_hostpool_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "hostpool_tsan")
    _build(["-fsanitize=thread"], exe)
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "failures 0" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_hostpool_stress_plain(tmp_path):
    exe = str(tmp_path / "hostpool_plain")
    _build(["-O2"], exe)
    r = subprocess.run([exe, "40000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failures 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
