"""tests/host/test_tracking_adapters: the TrackingInterface adapters (GPS_L1_CA_DLL_PLL_Tracking_HIP, Galileo_E1_DLL_PLL_VEML_Tracking_HIP,
GPS_L5_DLL_PLL_Tracking_HIP, Galileo_E5a_DLL_PLL_Tracking_HIP), the dll_pll_veml_tracking_hip GNU Radio block and the Dll_Pll_Conf ->
gsh_trk_conf mapper, compiled against the reference's own headers and checked against the reference's own tracking chain
(oracle/_ref/libgnsssdr_ref_trk.so).  The binary is built by __graft_entry__.build() where /root/reference is present and travels with the tree.

CPU: `conf` mode -- the mapper equals what dll_pll_veml_tracking's constructor derives, field by field, and the local replicas equal the block's.
GPU: both chains driven through general_work over the same stream, call for call (window positions, symbol timing, Prompt_I, Doppler, C/N0,
loss of lock, telemetry fault, unusable configurations)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_tracking_adapters")


def _bin():
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        if not g.build_tracking_adapter_test():
            pytest.skip("tests/host/test_tracking_adapters was not prebuilt and /root/reference is not present here")
    return BIN


def test_dll_pll_conf_mapping_equals_reference_constructor():
    r = subprocess.run([_bin(), "conf"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "TRACKING CONF OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_tracking_adapters_follow_the_reference_block(gpu):
    r = subprocess.run([_bin()], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout, r.stdout[-6000:] + r.stderr[-2000:]
