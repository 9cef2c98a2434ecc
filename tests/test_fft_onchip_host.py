"""CPU check of the on-chip FFT plans (gnss-sdr_amd/csrc/fft_onchip.h) without a GPU.

tests/host/libfft_onchip_host.so (built by __graft_entry__.build() from tests/host/fft_onchip_host.cc) executes the
SAME per-thread phase functions the HIP kernels of csrc/pcps_onchip.hip call -- register DFTs, inter-stage twiddles,
LDS address maps -- thread by thread on the host.  Compared with numpy's float64 FFT; the plan geometry is checked
against the LDS-bank rules stated in the header.
"""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "host", "libfft_onchip_host.so")

PLANS = [25000, 4000, 2048, 4096, 5000, 8000, 8192, 10000, 12500, 16000, 16384, 20000, 32768, 1000, 2000, 2500, 6250, 2046, 4092, 8184, 16368, 5456, 2560, 10240, 25600]


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: run python -c 'import __graft_entry__ as g; g.build()'")
    return C.CDLL(LIB)


@pytest.mark.parametrize("form", [1, 2], ids=["component_exchange", "phased_64bit_exchange"])
@pytest.mark.parametrize("n", PLANS)
def test_onchip_plan_matches_numpy_fft(host, n, form):
    """Both LDS exchange forms of every plan (the device picks one per plan, Plan::EX64; the other is one macro away in A/B builds)."""
    rng = np.random.default_rng(n)
    host.oc_host_fft_form.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    for trial in range(2):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        if trial == 1:
            x[:] = 0
            x[n // 3] = 1.0 + 0.5j  # an impulse exercises every twiddle
        out = np.zeros(n, np.complex64)
        assert host.oc_host_fft_form(n, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), form) == 0
        ref = np.fft.fft(x.astype(np.complex128))
        err = np.linalg.norm(out - ref) / np.linalg.norm(ref)
        assert err < 1.5e-6, (n, trial, err)
        assert np.max(np.abs(out - ref)) < 5e-6 * np.max(np.abs(ref)), (n, trial)


@pytest.mark.parametrize("n", PLANS)
def test_plan_geometry(host, n):
    info = (C.c_int * 8)()
    assert host.oc_host_plan_info(n, info) == 0
    threads, lds_floats, s1, s2, p2, ex64, np1, np2 = list(info)
    assert threads % 64 == 0 and threads <= 1024
    assert lds_floats * 4 + 1024 <= 160 * 1024          # the exchange buffer(s) of the form the plan uses + reduction scratch
    assert ex64 == (1 if (n * 4 > 72 * 1024 or threads == 1024) else 0)    # phased form where one work-group fills the compute unit anyway (fft_onchip.h: measured)
    assert 1 <= np1 <= 4 and 1 <= np2 <= 4
    assert p2 % 2 == 1                                   # exchange-2 writer lanes spread over the banks


def test_unknown_length_has_no_plan(host):
    out = np.zeros(8, np.float32)
    assert host.oc_host_fft(1234, out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == -1
