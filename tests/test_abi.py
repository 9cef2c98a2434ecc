"""CPU tests of the drop-in boundary: the C-ABI library builds, loads, exports exactly what include/gnss_sdr_hip.h
declares, and fails loudly (no CPU fallback) when no GPU is present.  No compute kernels run here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gnss_sdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gsh_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol(gsh):
    from gnss_sdr_amd import _lib
    declared = _header_functions()
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert _lib.missing_symbols() == []
    assert gsh.gsh_abi_version() == 25


def test_struct_layouts_match_header():
    from gnss_sdr_amd._lib import AcqConf, AcqResult, CorrJob
    assert C.sizeof(CorrJob) == 80
    assert CorrJob.shifts_chips.offset == 48
    assert C.sizeof(AcqResult) == 32
    assert C.sizeof(AcqConf) == 80


def test_no_gpu_means_loud_failure(gsh):
    if gsh.gsh_device_count() > 0:
        pytest.skip("a GPU is visible")
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.tracking import CorrelatorBank, HipMulticorrelatorRealCodes
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    with pytest.raises(GshError) as e:
        CorrelatorBank(1, 1023)
    assert e.value.code == 2  # GSH_ERR_NO_DEVICE
    with pytest.raises(GshError):
        HipMulticorrelatorRealCodes()
    with pytest.raises(GshError):
        PcpsAcquisitionBank(fs_in=4000000, fft_size=4000, doppler_max=5000, doppler_step=250, samples_per_chip=4, samples_per_code=4000.0)


def test_argument_validation_without_gpu(gsh):
    from gnss_sdr_amd._lib import AcqConf
    h = C.c_void_p()
    assert gsh.gsh_bank_create(0, 0, 1023, C.byref(h)) == 1          # GSH_ERR_INVALID: no slots
    assert gsh.gsh_bank_create(0, 1, 1 << 20, C.byref(h)) == 1       # code does not fit the LDS
    assert b"LDS" in gsh.gsh_last_error()
    c = AcqConf(fs_in=4000000, fft_size=4000, effective_fft_size=4000, consumed_samples=3000, doppler_max=5000,
                doppler_step=250, samples_per_chip=4, samples_per_code=4000.0, max_prn=1)
    assert gsh.gsh_acq_create(0, C.byref(c), C.byref(h)) == 1        # fft_size must be consumed or 2*consumed
    assert gsh.gsh_bank_launch(None, None) == 1
    assert gsh.gsh_mcorr_free(None) == 1


def test_compute_threshold_matches_boost_definition():
    """acq.cc:52-56 uses boost::math::gamma_p_inv; scipy.special.gammaincinv is the same function."""
    from gnss_sdr_amd.acquisition import compute_threshold
    from oracle.pcps_oracle import compute_threshold as ref_thr
    for pfa, eff, bins, dwells in [(0.001, 4000, 80, 1), (0.01, 25000, 41, 1), (1e-4, 128000, 41, 2), (0.1, 2048, 20, 4), (1e-6, 50000, 81, 8)]:
        assert compute_threshold(pfa, eff, bins, dwells) == pytest.approx(ref_thr(pfa, eff, bins, dwells), rel=2e-6), (pfa, eff, bins, dwells)


def test_acq_sizes_follow_reference_formulas():
    from gnss_sdr_amd.acquisition import acq_sizes
    s = acq_sizes(25000000)
    assert (s["consumed_samples"], s["fft_size"], s["effective_fft_size"], s["samples_per_chip"]) == (25000, 25000, 25000, 25)
    s = acq_sizes(4000000, bit_transition_flag=True)
    assert (s["consumed_samples"], s["fft_size"], s["effective_fft_size"], s["samples_per_chip"]) == (8000, 8000, 4000, 4)
    s = acq_sizes(32000000, sampled_ms=4, ms_per_code=4)
    assert (s["consumed_samples"], s["fft_size"]) == (128000, 128000)
    s = acq_sizes(4000000, sampled_ms=4, ms_per_code=1)
    assert (s["consumed_samples"], s["fft_size"]) == (16000, 32000)
