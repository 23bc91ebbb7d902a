import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The probe kernel (ANDs of a posting-stream cover and bitmap operands) serves shards of 40 M docs and more by default; the parity corpora are
# smaller, so the tests move the line down (read once, when the library first compiles a request).  The default routing of small shards —
# k_scan_simple — is what the VQ_NO_PROBE leg of test_alternative_kernel_routes_match runs.
os.environ.setdefault("VQ_PROBE_MIN_DOCS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
