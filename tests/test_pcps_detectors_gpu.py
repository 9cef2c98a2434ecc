"""GPU parity tests of the Tong, Galileo 8 ms, CCCWSR, QuickSync and fine-Doppler detectors (gnss_sdr_amd/detectors.py over the C ABI:
weighted grid accumulation, input power, per-bin peaks) against oracle/pcps_oracle.py (TongOracle, Galileo8msOracle, CccwsrOracle,
QuickSyncOracle, FineDopplerOracle) on the reference's own
synthetic cases.  Bars: the state / counter trajectory, peak time index and Doppler bin of every dwell equal the oracle's
(integer work: bit-exact); input power within 1e-6 relative (float sum order, see gsh_acq_input_power); statistics within
RTOL (float32 transforms of different factorisation)."""
import numpy as np
import pytest

from oracle.pcps_oracle import CccwsrOracle, FineDopplerOracle, Galileo8msOracle, QuickSyncOracle, TongOracle
from detector_cases import cccwsr_case, e1_8ms_case, fine_doppler_case, quicksync_case, tong_case

pytestmark = pytest.mark.gpu

RTOL = 2e-3
# QuickSync wipes off blocks of up to 32 000 samples with the reference's table kernel (volk_gnsssdr_s32f_sincos_32fc, pinned bit-exact
# in the oracle), whose fixed-point phase drifts up to 0.043 rad from exp(-j 2 pi f n / fs) by the end of such a block (0.005 rad at
# 750 Hz).  The engine's phasor is exact, so values agree to the reference's own table error, not to float rounding: a signal peak moves
# by ~1e-3, the maximum of a noise-only grid by up to ~1e-2.  Indices, aliases, states are compared exactly.
RTOL_QS = 5e-3
RTOL_QS_NOISE = 2e-2


@pytest.mark.parametrize("path", [0, 1])
@pytest.mark.parametrize("signal", [True, False])
def test_tong_trajectory_matches_oracle(gpu, path, signal):
    from gnss_sdr_amd.detectors import PcpsTongAcquisition
    x, kw, code = tong_case(signal=signal, seed=2013 if signal else 5)
    if not signal:
        kw = dict(kw, threshold=1e-5, tong_max_val=50, tong_max_dwells=5)  # counts up on noise until tong_max_dwells
    o = TongOracle(**kw)
    g = PcpsTongAcquisition(device=gpu, transform_path=path, **kw)
    assert g.n_bins == o.n_bins == 81
    o.set_local_code(code)
    g.set_local_code(code)
    k = 0
    while o.state == 1:
        blk = x[k * 4000:(k + 1) * 4000]
        so, sg = o.work(blk), g.work(blk)
        assert (sg, g.tong_count, g.dwell_count) == (so, o.tong_count, o.dwell_count)
        assert abs(float(g.input_power) - float(o.input_power)) <= 1e-6 * float(o.input_power)
        assert abs(float(g.weight) - float(o.weight)) <= 2e-6 * float(o.weight)
        assert abs(float(g.mag) - float(o.mag)) <= RTOL * float(o.mag)
        if signal:
            assert (g.result["index_time"], g.result["index_doppler"]) == (o.result["index_time"], o.result["index_doppler"])
            assert g.result["acq_delay_samples"] == o.result["acq_delay_samples"] and g.result["doppler_hz"] == o.result["doppler_hz"]
        k += 1
    assert g.state == o.state == (2 if signal else 3)
    # the accumulated grid itself (d_grid_data, tong.cc:249)
    grid = g.bank.read_grid(0)
    assert np.max(np.abs(grid - o.grid)) <= RTOL * float(np.max(o.grid))
    # a second run after init() starts from an empty grid (tong.cc:176-182)
    g.init()
    o.init()
    assert g.work(x[:4000]) == o.work(x[:4000]) and abs(float(g.mag) - float(o.mag)) <= RTOL * float(o.mag)
    g.close()


def test_grid_weight_rules(gpu):
    from gnss_sdr_amd._lib import GshError
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    b = PcpsAcquisitionBank(4000000, 4000, 5000, 500, 4, 4000.0, device=gpu, keep_grid=False)
    with pytest.raises(GshError):
        b.set_grid_weight(0.5)     # no stored grid to weight
    b.set_grid_weight(1.0)
    with pytest.raises(GshError):
        b.input_power()            # nothing resident yet
    with pytest.raises(GshError):
        b.dwell_resident(1)
    b.close()
    b = PcpsAcquisitionBank(4000000, 4000, 5000, 500, 4, 4000.0, device=gpu)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(4000) + 1j * rng.standard_normal(4000)).astype(np.complex64)
    b.set_local_code(0, np.sign(rng.standard_normal(4000)).astype(np.complex64))
    r1 = b.dwell(x, 1)[0]
    g1 = b.read_grid(0)
    b.set_grid_weight(0.25)        # a power of two: every weighted cell is exactly a quarter
    b.stage_input(x)
    r2 = b.dwell_resident(1)[0]
    assert np.array_equal(b.read_grid(0), g1 * np.float32(0.25))
    assert (r2["index_time"], r2["index_doppler"]) == (r1["index_time"], r1["index_doppler"]) and r2["peak"] == r1["peak"] * 0.25
    pk, ix = b.read_row_peaks(0)
    assert np.array_equal(pk, g1.max(axis=1) * np.float32(0.25)) and np.array_equal(ix, g1.argmax(axis=1).astype(np.uint32))
    p = b.input_power()
    assert abs(p - float(np.mean(np.abs(x.astype(np.complex128)) ** 2))) < 1e-6 * p
    b.close()


@pytest.mark.parametrize("flip", [False, True])
def test_8ms_matches_oracle(gpu, flip):
    from gnss_sdr_amd.detectors import GalileoPcps8msAcquisition
    x, kw, code, _ = e1_8ms_case(flip)
    o = Galileo8msOracle(**kw)
    g = GalileoPcps8msAcquisition(device=gpu, **kw)
    o.set_local_code(code)
    g.set_local_code(code)
    so, sg = o.work(x[:32000]), g.work(x[:32000])
    assert sg == so == 2
    # code A is two identical periods, so its correlation is exactly 16000-periodic: y[t] == y[t + 16000] up to rounding, and
    # which twin a float32 transform ranks first is not defined.  The block reports indext % samples_per_code (8ms.cc:255).
    fold = lambda r: dict(r, index_time=r["index_time"] % 16000)
    assert fold(g.result) == fold(o.result)
    assert abs(float(g.input_power) - float(o.input_power)) <= 1e-6 * float(o.input_power)
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL * float(o.test_statistics)
    d = o.result["index_doppler"]
    for k in (0, 2):  # per-bin maxima of code A and code B around the winning bin
        assert abs(g.rows[d][k] - o.rows[d][k]) <= RTOL * o.rows[d][k]
    assert (g.rows[d][1], g.rows[d][3])[o.result["code"]] % 16000 == (o.rows[d][1], o.rows[d][3])[o.result["code"]] % 16000
    g.close()


def test_8ms_noise_only_negative(gpu):
    from gnss_sdr_amd.detectors import GalileoPcps8msAcquisition
    x, kw, code, _ = e1_8ms_case(False, signal=False)
    o = Galileo8msOracle(**kw)
    g = GalileoPcps8msAcquisition(device=gpu, **kw)
    o.set_local_code(code)
    g.set_local_code(code)
    assert g.work(x[:32000]) == o.work(x[:32000]) == 3
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL * float(o.test_statistics)
    g.close()


@pytest.mark.parametrize("fs,p,path", [(8000000, 4, 0), (4000000, 4, 0), (8000000, 2, 0), (8000000, 2, 1)])
def test_quicksync_matches_oracle(gpu, fs, p, path):
    """Folded search + alias resolution.  (8 Msps, p = 2) folds into the 4000-point on-chip plan (path 0) and is repeated through
    the four-step kernels (path 1); the other lengths (2000, 1000) have no plan and take the four-step path either way."""
    from gnss_sdr_amd.detectors import PcpsQuickSyncAcquisition
    x, kw, code = quicksync_case(fs, p)
    o = QuickSyncOracle(**kw)
    g = PcpsQuickSyncAcquisition(device=gpu, transform_path=path, **kw)
    assert (g.fft_size, g.n_in, g.n_bins) == (o.fft_size, o.n_in, o.n_bins)
    o.set_local_code(code)
    g.set_local_code(code)
    assert g.work(x) == o.work(x) == 2
    assert g.result == o.result
    assert abs(float(g.input_power) - float(o.input_power)) <= 1e-6 * float(o.input_power)
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL_QS * float(o.test_statistics)
    # the alias correlations themselves (sequential float sums in the reference, double sums here)
    assert np.max(np.abs(g.candidates - o.candidates)) <= RTOL_QS * np.max(np.abs(o.candidates))
    # per-bin maxima of the folded grid around the winner
    d = o.result["index_doppler"]
    for dd in (d - 1, d, d + 1):
        assert abs(g.rows[dd][0] - o.rows[dd][0]) <= RTOL_QS * o.rows[d][0]
    assert g.rows[d][1] == o.rows[d][1]
    g.close()


def test_quicksync_noise_only_and_bit_transition_rule(gpu):
    from gnss_sdr_amd.detectors import PcpsQuickSyncAcquisition
    x, kw, code = quicksync_case(signal=False, seed=3)
    o = QuickSyncOracle(**kw)
    g = PcpsQuickSyncAcquisition(device=gpu, **kw)
    o.set_local_code(code)
    g.set_local_code(code)
    assert g.work(x) == o.work(x) == 3
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL_QS_NOISE * float(o.test_statistics)
    g.close()
    # bit_transition_flag: two dwells, decided only after the second one (qs.cc:377-390)
    xs, kw, code = quicksync_case()
    g = PcpsQuickSyncAcquisition(device=gpu, bit_transition_flag=True, **kw)
    g.set_local_code(code)
    assert g.max_dwells == 2 and g.work(xs) == 1 and g.work(xs) == 2
    g.close()


def test_fold_conf_rules(gpu):
    from gnss_sdr_amd._lib import GshError
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    with pytest.raises(GshError):   # consumed_samples must be fold * fft_size
        PcpsAcquisitionBank(4000000, 1000, 5000, 500, 1, 4000.0, consumed_samples=4000, fold=16, device=gpu)
    with pytest.raises(GshError):   # no bit-transition placement with folding
        PcpsAcquisitionBank(4000000, 1000, 5000, 500, 1, 4000.0, consumed_samples=16000, effective_fft_size=500, fold=16, bit_transition_flag=True, device=gpu)
    b = PcpsAcquisitionBank(4000000, 1000, 5000, 500, 1, 4000.0, consumed_samples=16000, fold=16, device=gpu)
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(16000) + 1j * rng.standard_normal(16000)).astype(np.complex64)
    b.set_local_code(0, np.sign(rng.standard_normal(1000)).astype(np.complex64))
    b.dwell(x, 1)
    with pytest.raises(GshError):   # a candidate window must stay inside the resident block
        b.time_correlate(np.ones(4000, np.complex64), 0, [12001])
    with pytest.raises(GshError):
        b.time_correlate(np.ones(4000, np.complex64), 99, [0])
    # bin 10 is 0 Hz here (-5000 + 10 * 500): the correlation is a plain dot product
    got = b.time_correlate(np.ones(4000, np.complex64), 10, [0, 12000])
    ref = np.array([x[:4000].astype(np.complex128).sum(), x[12000:].astype(np.complex128).sum()])
    assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref)) + 1e-4
    b.close()


@pytest.mark.parametrize("consistent", [False, True])
@pytest.mark.parametrize("fs", [4000000, 25000000])
def test_fine_doppler_matches_oracle(gpu, consistent, fs):
    """Accumulated grid + peak-ratio decision + zero-padded fine-Doppler transform (320 000 points at 4 Msps, 2 000 000 at 25 Msps):
    states, delay, grid Doppler and the fine transform's arg-max equal the oracle's."""
    from gnss_sdr_amd.detectors import PcpsAcquisitionFineDoppler
    x, kw, code = fine_doppler_case(fs=fs)
    n = fs // 1000
    o = FineDopplerOracle(consistent_grid=consistent, **kw)
    g = PcpsAcquisitionFineDoppler(consistent_grid=consistent, device=gpu, **kw)
    o.set_local_code(code)
    g.set_local_code(code)
    for k in range(2):
        assert g.dwell(x[k * n:(k + 1) * n]) == o.dwell(x[k * n:(k + 1) * n])
    assert g.decide() == o.decide() == 3
    assert g.result == o.result
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL * float(o.test_statistics)
    assert g.estimate_doppler(x[2 * n:]) == o.estimate_doppler(x[2 * n:]) == 4
    assert g.fine_index == o.fine_index and g.fine_doppler == o.fine_doppler and g.result == o.result
    assert abs(g.fine_doppler - 1730.0) <= fs / (80.0 * n) + 1e-6
    g.close()


def test_spectrum_peak_primitive(gpu):
    import ctypes as C
    from gnss_sdr_amd import _lib
    from gnss_sdr_amd._lib import GshError, check, fptr
    lib = _lib.load()
    rng = np.random.default_rng(9)
    n, M = 3000, 24000
    x = (0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) + np.exp(2j * np.pi * (-5000.5 / M) * np.arange(n))).astype(np.complex64)
    k, pk = C.c_uint32(0), C.c_float(0.0)
    check(lib.gsh_spectrum_peak(gpu, fptr(x), None, n, M, C.byref(k), C.byref(pk)))
    z = np.zeros(M, np.complex128)
    z[:n] = x
    Z = np.abs(np.fft.fft(z)) ** 2
    assert int(k.value) == int(np.argmax(Z)) and abs(pk.value - Z.max()) <= 1e-4 * Z.max()
    # equal maxima: a real-valued input has |X[k]| = |X[M - k]|; the lower index wins (volk_gnsssdr_32f_index_max_32u)
    xr = np.cos(2 * np.pi * 100.0 / 1000 * np.arange(1000)).astype(np.complex64)
    check(lib.gsh_spectrum_peak(gpu, fptr(xr), None, 1000, 1000, C.byref(k), C.byref(pk)))
    assert int(k.value) == 100
    with pytest.raises(GshError):
        check(lib.gsh_spectrum_peak(gpu, fptr(x), None, n, 2 * 1009 * 1013, C.byref(k), C.byref(pk)))   # no four-step split


def _cccwsr_pair(gpu, kw, cd, cp, path=0):
    from gnss_sdr_amd.detectors import PcpsCccwsrAcquisition
    o = CccwsrOracle(**kw)
    g = PcpsCccwsrAcquisition(device=gpu, transform_path=path, **kw)
    o.set_local_code(cd, cp)
    g.set_local_code(cd, cp)
    return o, g


def _cccwsr_compare_published(o, g):
    """what the block publishes: state, delay, Doppler, statistic (+ the input power behind it)"""
    assert g.state == o.state
    for k in ("acq_delay_samples", "doppler_hz", "doppler_step", "index_time", "index_doppler"):
        assert g.result[k] == o.result[k], k
    assert abs(float(g.input_power) - float(o.input_power)) <= 1e-6 * float(o.input_power)
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL * float(o.test_statistics)


@pytest.mark.parametrize("path", [0, 1])
@pytest.mark.parametrize("ds,ps", [(1.0, -1.0), (1.0, 1.0), (-1.0, 1.0), (-1.0, -1.0)])
def test_cccwsr_quadrature_matches_oracle(gpu, ds, ps, path):
    """pilot in quadrature: one branch combines coherently, the other cancels -- the branch is defined and must equal the oracle's,
    which forms the two branches the block's way (two inverse transforms + element-wise +-j combination, cccwsr.cc:212-252)
    while the engine correlates once per branch with the combined local codes."""
    x, kw, cd, cp, _ = cccwsr_case("quadrature", data_sign=ds, pilot_sign=ps)
    o, g = _cccwsr_pair(gpu, kw, cd, cp, path)
    assert o.work(x[:16000]) == g.work(x[:16000]) == 2
    _cccwsr_compare_published(o, g)
    assert g.result["branch"] == o.result["branch"]
    d = o.result["index_doppler"]
    big = max(o.rows[d][0], o.rows[d][2])
    for k in (0, 2):   # both branch maxima at the winning bin; the cancelled branch is noise-level, so scale by the winner
        assert abs(g.rows[d][k] - o.rows[d][k]) <= RTOL * big
    w = 1 + 2 * o.result["branch"]
    assert g.rows[d][w] == o.rows[d][w]
    g.close()


def test_cccwsr_reference_case_inphase(gpu):
    """galileo_e1_pcps_cccwsr_ambiguous_acquisition_gsoc2013_test.cc config_2: E1B - E1C on one carrier phase.  Both branches then
    peak at the same cell with nearly equal values, and which one the reference itself picks is decided by float rounding
    (oracle/pcps_oracle.py CccwsrOracle note): the published values are compared, the branch is not."""
    x, kw, cd, cp, delay = cccwsr_case("inphase")
    o, g = _cccwsr_pair(gpu, kw, cd, cp)
    assert o.work(x[:16000]) == g.work(x[:16000]) == 2
    _cccwsr_compare_published(o, g)
    d = o.result["index_doppler"]
    assert g.rows[d][1] == g.rows[d][3] == o.rows[d][1] == o.rows[d][3]
    for k in (0, 2):
        assert abs(g.rows[d][k] - o.rows[d][k]) <= RTOL * o.rows[d][k]
    g.close()


def test_cccwsr_noise_only_and_running_maximum(gpu):
    x, kw, cd, cp, _ = cccwsr_case(signal=False)
    o, g = _cccwsr_pair(gpu, kw, cd, cp)
    assert o.work(x[:16000]) == g.work(x[:16000]) == 3
    assert abs(float(g.test_statistics) - float(o.test_statistics)) <= RTOL * float(o.test_statistics)
    g.close()
    # two dwells of one acquisition: the peak of dwell 1 survives dwell 2 (cccwsr.cc:160), the power is dwell 2's (cccwsr.cc:192-194)
    xs, kw, cd, cp, _ = cccwsr_case("quadrature", n_blocks=1)
    o, g = _cccwsr_pair(gpu, dict(kw, max_dwells=2, threshold=1e9), cd, cp)
    assert o.work(xs[:16000]) == g.work(xs[:16000]) == 1
    mag1 = g.mag
    assert o.work(x[:16000]) == g.work(x[:16000]) == 3
    assert g.mag == mag1
    _cccwsr_compare_published(o, g)
    g.close()


E5A_GPU_CASES = {
    # name: (case kwargs, transform path): on-chip plans (16 000, 25 000), the split plan of 32 000 points and the four-step kernels (24 000, 36 000, 20 480)
    "cfg1_32Msps_1ms_split": (dict(fs=32000000, sampled_ms=1, doppler=2800.0, delay_chips=4475.0, doppler_max=10000, cn0=50.0), 0),
    "cfg2_12Msps_3ms_fourstep": (dict(fs=12000000, sampled_ms=3), 0),
    "3ms_data_flip_first": (dict(fs=8000000, sampled_ms=3, data_signs=(-1, 1, 1), pilot_signs=(1, 1, 1), delay_chips=10.0), 0),
    # (no 2 ms case with both components AND a sign change: a 2 ms B code [-c, c] is exactly antiperiodic, so its |.|^2 row has two EXACTLY equal
    #  peaks one period apart whose ranking is FFT rounding noise -- harmless by itself (same delay modulo the code period), but :393 then ranks Q-B by
    #  the I-B row READ AT that arg-max, and I-B = [-I, Q] is not antiperiodic: the reference's own choice is rounding noise there, measured with
    #  profiles/ab/dbg_e5a.py)
    "3ms_both_flip_onchip": (dict(fs=5456000, sampled_ms=3, data_signs=(-1, 1, 1), pilot_signs=(-1, 1, 1), delay_chips=10.0, doppler=-1300.0), 0),
    "3ms_pilot_flip_onchip": (dict(fs=5456000, sampled_ms=3, data_signs=(1, 1, 1), pilot_signs=(-1, 1, 1), delay_chips=7.0, doppler=900.0), 0),
    "3ms_pilot_flip_fourstep": (dict(fs=5456000, sampled_ms=3, data_signs=(1, 1, 1), pilot_signs=(-1, 1, 1), delay_chips=7.0, doppler=900.0), 1),
    "2ms_data_only_flip_onchip": (dict(fs=12500000, sampled_ms=2, both=False, data_signs=(-1, 1), delay_chips=7.0, doppler=900.0), 0),
    "3ms_data_only": (dict(fs=8000000, sampled_ms=3, both=False, data_signs=(-1, 1, 1), delay_chips=10.0), 0),
    "1ms_data_only_onchip": (dict(fs=10240000, sampled_ms=1, both=False, cn0=50.0), 0),
    "2ms_caf_onchip": (dict(fs=8000000, sampled_ms=2, caf_window_hz=1500, doppler=-2100.0), 0),
    "1ms_caf_data_only": (dict(fs=10240000, sampled_ms=1, both=False, caf_window_hz=1000, cn0=50.0, doppler=4900.0), 0),
    "zero_padding": (dict(fs=8000000, sampled_ms=2, zero_padding=1, cn0=50.0), 0),
    "noise_only_2_dwells": (dict(fs=5456000, sampled_ms=3, signal=False, max_dwells=2, n_blocks=3), 0),
}


@pytest.mark.parametrize("name", list(E5A_GPU_CASES))
def test_e5a_noncoherent_iq_matches_oracle(gpu, name):
    """Galileo E5a non-coherent I + Q search (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc) on the device engine against E5aNoncoherentIqOracle, which is
    pinned to the reference block itself (tests/test_pcps_oracle_pinned.py): per Doppler bin the same A / B choice for both components, the same arg-max of
    the SUM of the two kept magnitude rows and the same CAF inputs; state, delay, Doppler (after the CAF filter) equal, values within RTOL."""
    from gnss_sdr_amd.detectors import GalileoE5aNoncoherentIQAcquisitionCaf
    from oracle.pcps_oracle import E5aNoncoherentIqOracle
    from detector_cases import e5a_case
    case, path = E5A_GPU_CASES[name]
    x, kw, ci, cq = e5a_case(**case)
    if name.startswith("noise"):
        kw["threshold"] *= 3.0
    n = kw["fft_size"]
    o = E5aNoncoherentIqOracle(**kw)
    g = GalileoE5aNoncoherentIQAcquisitionCaf(device=gpu, transform_path=path, **kw)
    assert g.n_bins == o.n_bins and g.sampled_ms == o.sampled_ms
    o.set_local_code(ci, cq)
    g.set_local_code(ci, cq)
    fnf = np.float32(n) * np.float32(n)
    for dwell in range(kw["max_dwells"]):
        blk = x[dwell * n:(dwell + 1) * n]
        so, sg = o.work(blk), g.work(blk)
        slot_name = {v: k for k, v in g.slots.items()}
        decided = 0
        for d in range(o.n_bins):
            sel_i, sel_q, peak, t = o.rows[d]
            r = g.rows[d]
            # in a noise bin the A and B row maxima of a component are two unrelated noise peaks; where they happen to lie within float32 FFT
            # rounding of each other the two engines may rank them differently, so the per-bin values are held where the choices agree (and
            # most bins must), while the winning bin's choice and indices are asserted unconditionally below
            if slot_name[int(r["i_slot"])] == sel_i and (sel_q is None or slot_name[int(r["q_slot"])] == sel_q):
                decided += 1
                # values: the oracle wipes off with the reference's table kernel, which has drifted from the true phase by the end of a 16 000 - 36 000
                # sample block (see RTOL_QS above); the engine's phasor is exact
                assert float(r["peak"]) == pytest.approx(peak, rel=RTOL_QS_NOISE), (name, d)
                assert float(r["caf_i"]) == pytest.approx(float(o.caf_i[d]), rel=RTOL_QS_NOISE)
                if sel_q is not None:
                    assert float(r["caf_q"]) == pytest.approx(float(o.caf_q[d]), rel=RTOL_QS_NOISE)
        assert decided >= 0.8 * o.n_bins, (name, decided, o.n_bins)
        assert sg == so and g.well_count == o.well_count
        assert float(g.input_power) == pytest.approx(float(o.input_power), rel=1e-6)
        tol = RTOL_QS if case.get("signal", True) else RTOL_QS_NOISE
        assert float(g.mag) == pytest.approx(float(o.mag), rel=tol)
        assert float(g.test_statistics) == pytest.approx(float(o.test_statistics), rel=tol)
        if case.get("signal", True):
            dg, do = g.result, o.result
            # (index_time modulo the code period: a block of several identical code periods has that many equal peaks; the block publishes the remainder)
            assert (dg["index_doppler"], dg["index_time"] % kw["samples_per_code"], dg["acq_delay_samples"], dg["doppler_hz"]) == \
                   (do["index_doppler"], do["index_time"] % kw["samples_per_code"], do["acq_delay_samples"], do["doppler_hz"]), (name, dg, do)
            wi = o.rows[do["index_doppler"]]
            wr = g.rows[do["index_doppler"]]
            assert (slot_name[int(wr["i_slot"])], None if wi[1] is None else slot_name[int(wr["q_slot"])]) == (wi[0], wi[1])
    assert o.state == (4 if name.startswith("noise") else 3)
    g.close()


def test_pair_peaks_argument_rules(gpu):
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    n = 8000
    x = (np.random.default_rng(1).standard_normal(n) + 1j * np.random.default_rng(2).standard_normal(n)).astype(np.complex64)
    code = np.sign(np.random.default_rng(3).standard_normal(n)).astype(np.complex64)
    b = PcpsAcquisitionBank(8000000, n, 1000, 250, 1, 8000.0, max_prn=2, device=gpu)
    b.set_local_code(0, code)
    b.set_local_code(1, -code)
    with pytest.raises(GshError):
        b.noncoherent_pair_peaks(0, 1)                 # no dwell yet
    b.dwell(x, 2)
    pk = b.noncoherent_pair_peaks(0, 1)
    g0, g1 = b.read_grid(0), b.read_grid(1)
    s = (g0 + g1).astype(np.float32)
    assert np.array_equal(pk["peak"], s.max(axis=1)) and np.array_equal(pk["index_time"], np.argmax(s, axis=1))   # bit-exact: one float add per cell
    assert np.array_equal(pk["caf_i"], g0.max(axis=1)) and np.array_equal(pk["caf_q"], g1.max(axis=1))
    one = b.noncoherent_pair_peaks(1)                  # data only: the row itself
    assert np.array_equal(one["peak"], g1.max(axis=1)) and np.array_equal(one["index_time"], np.argmax(g1, axis=1))
    for bad in ((2, -1, -1, -1), (0, 2, -1, -1), (0, -1, -1, 1), (0, 1, -1, 1)):
        with pytest.raises(GshError):
            b.noncoherent_pair_peaks(*bad)
    b.close()
    nog = PcpsAcquisitionBank(8000000, n, 1000, 250, 1, 8000.0, max_prn=2, device=gpu, keep_grid=False)
    nog.set_local_code(0, code)
    nog.set_local_code(1, code)
    nog.dwell(x, 2)
    with pytest.raises(GshError):
        nog.noncoherent_pair_peaks(0, 1)               # the rows to be added are not stored
    nog.close()
