"""Runs the C++ host-side test program (tests/host/test_host_classes.cc) on the GPU box: the reference-shaped C++
classes of gnss-sdr_amd/host/ driven with the reference's call pattern, checked against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_host_classes")


@pytest.mark.gpu
def test_cpp_host_classes(gpu):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_host_test()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HOST CLASSES OK" in r.stdout, r.stdout + r.stderr
