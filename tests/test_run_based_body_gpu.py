"""The experimental run-based correlator body (csrc/mcorr_device.h run_segment_runs; DESIGN section 3, profiles/r02/run_based_experiment.txt) is selected by
GSH_MC_PACKED_BODY=2, which the library reads once per process -- so its parity run is a child process: the chip-selection (bit-exact), reference unit-test,
config-2 and edge-case tests of tests/test_tracking_gpu.py with the path forced.  It is slower than the packed trips and not the product path; the test keeps it honest
while it stays in the tree.  Round 4: its two launch flavours carry 16 - 32 B of scratch per thread, so the regular library no longer holds them (-DGSH_MC_RUNS_EXPERIMENT,
profiles/ab/build_variant.py); the test runs against a library that was built with the macro (GSH_LIB_PATH) and skips otherwise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_run_based_body_passes_the_tracking_parity_tests(gpu):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import gnss_sdr_amd
    from kernel_metadata import kernels
    if not any("mcorr_kernelILi3ELi0ELb0ELb1E" in n for n in kernels(gnss_sdr_amd._lib.LIB_PATH)):
        pytest.skip("library built without -DGSH_MC_RUNS_EXPERIMENT")
    env = dict(os.environ, GSH_MC_PACKED_BODY="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_tracking_gpu.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
