"""Runs tests/host/test_adapters on the GPU box: the AcquisitionInterface adapters of gnss-sdr_amd/host/gnss_sdr_adapters/ (GPS L1 C/A,
Galileo E1, GPS L5; gr_complex and cshort items; make_two_steps) built against the reference's own interface headers and driven
through the GNU Radio general_work contract until they publish their "events" message.  The binary is built by
__graft_entry__.build() in the container that has /root/reference (it cannot be built on the GPU box) and travels with the tree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_adapters")


@pytest.mark.gpu
def test_gnss_sdr_adapters_end_to_end(gpu):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        if not g.build_adapter_test():
            pytest.skip("tests/host/test_adapters was not prebuilt and /root/reference is not present here")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ADAPTERS OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
