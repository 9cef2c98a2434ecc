import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def gsh():
    """The built C-ABI library (fails loudly when it has not been built)."""
    try:
        # tests that also use torch for device memory: torch must bring in ITS HIP runtime before libgnss_sdr_hip.so pulls in
        # /opt/rocm's (two runtimes in one process: the second one to initialise sees no GPU).  bench.py has the same order.
        import torch  # noqa: F401
    except ImportError:
        pass
    import gnss_sdr_amd
    return gnss_sdr_amd.load()


@pytest.fixture(scope="session")
def gpu(gsh):
    n = gsh.gsh_device_count()
    if n < 1:
        pytest.fail("a test marked gpu ran on a machine without a visible HIP device")
    return 0
