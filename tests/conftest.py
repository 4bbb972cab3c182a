import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _resource_trace(request):
    """ADCRAFT_TEST_TRACE=<file>: one line per test with the process's open descriptors and resident memory after it -
    what leaked engines (streams, page-locked buffers) show up in first"""
    yield
    path = os.environ.get("ADCRAFT_TEST_TRACE")
    if path:
        with open("/proc/self/statm") as f:
            rss_mb = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") >> 20
        with open(path, "a") as f:
            with open("/proc/self/maps") as m:
                libs = sorted({line.split()[-1] for line in m if ".so" in line and ("torch/lib" in line or "rccl" in line or "hsa-runtime" in line)})
            f.write(f"{request.node.nodeid} fds={len(os.listdir('/proc/self/fd'))} rss_mb={rss_mb} torch={'torch' in sys.modules} {libs}\n")
