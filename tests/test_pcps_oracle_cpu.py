"""CPU tests of the acquisition oracle (oracle/pcps_oracle.py): the reference's synthetic known-answer case and the
agreement of its float32 path with the float64 arbiter.  (The GPU engine is compared with this oracle in
tests/test_acquisition_gpu.py.)"""
import numpy as np
import pytest

import oracle
from oracle.pcps_oracle import PcpsOracle, compute_threshold
from helpers import synth_gps_l1_stream


def test_gsoc2013_known_answer_on_the_oracle():
    """gps_l1_ca_pcps_acquisition_gsoc2013_test.cc:197-264,384-401: PRN 10, 750 Hz, 600 chips, 4 Msps,
    doppler_max 10000, step 250, pfa 1e-3: delay error < 0.5 chip, Doppler error < 2/(3 T)."""
    fs, n = 4000000, 4000
    x = synth_gps_l1_stream(n, fs, [10], [750.0], [1023.0 - 600.0], cn0_dbhz=47.0, seed_noise=2013)
    kw = dict(fs_in=fs, fft_size=n, doppler_max=10000, doppler_step=250, samples_per_chip=4, samples_per_code=4000.0)
    for use_cfar in (True, False):
        o = PcpsOracle(use_cfar=use_cfar, **kw)
        assert o.n_bins == 80
        o.set_local_code(oracle.ca_code_complex_sampled(10, fs))
        r = o.dwell(x)
        assert abs(600.0 - r["acq_delay_samples"] * 1023.0 / 4000.0) < 0.5
        assert abs(r["doppler_hz"] - 750.0) < 2.0 / 3e-3
        if use_cfar:
            assert r["test_statistics"] > compute_threshold(0.001, n, 80, 1)
        else:
            assert r["test_statistics"] > 2.0
        p = PcpsOracle(use_cfar=use_cfar, precise=True, **kw)
        p.set_local_code(oracle.ca_code_complex_sampled(10, fs))
        rp = p.dwell(x)
        assert (r["index_time"], r["index_doppler"]) == (rp["index_time"], rp["index_doppler"])
        assert abs(r["test_statistics"] - rp["test_statistics"]) <= 2e-3 * rp["test_statistics"]


def test_exclusion_window_wraps_like_the_reference():
    """first_vs_second_peak_statistic blanks [tau-spc, tau+spc) cyclically (acq.cc:485-509), including across the ends."""
    kw = dict(fs_in=4000000, fft_size=4000, doppler_max=1000, doppler_step=500, samples_per_chip=4, samples_per_code=4000.0, use_cfar=False)
    o = PcpsOracle(**kw)
    g = np.ones((o.n_bins, 4000), np.float32)
    g[1, 2] = 50.0      # peak near the start: window wraps to the end of the row
    g[1, 3998] = 40.0   # inside the wrapped window -> must be blanked
    g[1, 5] = 30.0      # inside the window (idx 2-4 .. 2+4 exclusive)
    g[1, 6] = 20.0      # first cell after the window -> the second peak
    o.grid = g
    r = o.statistics()
    assert (r["index_time"], r["index_doppler"]) == (2, 1)
    assert r["second_peak"] == 20.0 and r["test_statistics"] == 2.5


def test_oracle_step_two_refines_doppler():
    """make_two_steps (acq.cc:294-301, 428-437, 609-624): the narrow grid sits at center + (d - floor(nb2/2)) * step2 in float32,
    the reported Doppler is the truncated float, and the CFAR statistic divides by step one's input power."""
    from helpers import synth_gps_l1_stream
    import oracle
    from oracle.pcps_oracle import PcpsOracle
    fs, n = 4000000, 4000
    x = synth_gps_l1_stream(2 * n, fs, [6], [1437.0], [321.4], cn0_dbhz=50.0, seed_noise=1)
    o = PcpsOracle(fs, n, 5000, 250, 4, float(n))
    o.set_local_code(oracle.ca_code_complex_sampled(6, fs))
    r1 = o.dwell(x[:n])
    assert r1["doppler_hz"] in (1250, 1500)
    r2 = o.dwell_step2(x[n:], float(r1["doppler_hz"]), 4, 125.0, input_power_step_one=r1["input_power"])
    assert r2["freqs"] == [r1["doppler_hz"] + k * 125.0 for k in (-2, -1, 0, 1)]
    assert abs(r2["doppler_hz"] - 1437.0) <= 125.0
    assert r2["test_statistics"] == pytest.approx(r2["peak"] / r1["input_power"], rel=1e-6)
    assert o.n_bins == 40 and o.grid.shape == (40, n)      # the wide-grid state is untouched
    r3 = o.dwell_step2(x[n:], 1437.4, 5, 62.5)
    assert r3["freqs"][2] == float(np.float32(1437.4)) and r3["doppler_hz"] == int(np.float32(np.float32(1437.4) + (np.float32(r3["index_doppler"]) - np.float32(2)) * np.float32(62.5)))
