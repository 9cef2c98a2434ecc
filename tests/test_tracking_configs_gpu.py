"""GPU parity tests of the multicorrelator at the shapes of BASELINE configs 4 and 5.

Config 4: Galileo E1, 50 channels, fs 32 Msps -> N = 128 000 per 4 ms code period, 5-tap VE/E/P/L/VL on the pilot
          (E1C) replica + the 1-tap data (E1B) correlator of track_pilot mode (trk.cc:1246-1256), local replica
          sinBOC(1,1) at 2 samples per chip = 8184 floats (trk.cc:289, 837-844), taps in code samples
          {-0.5, -0.15, 0, +0.15, +0.5} chips x 2 (trk.cc:632-636).
Config 5: 256 channels at fs 50 Msps in ONE launch: 96 GPS L1 (N 50 000, 3 taps, 1023-chip code), 96 Galileo E1
          (N 200 000, 5 + 1 taps, 8184), 64 GPS L5 (N 50 000, 3 taps, 10 230 chips), two RF streams.

Replicas come from tests/golden/codes_e1_l5.npz (minted from the reference's own generators).  Same bars as
tests/test_tracking_gpu.py.  N >= 65 536 also exercises the reference's `unsigned n*n` wrap in the high-dynamics
resampler (K/volk_gnsssdr_32f_xn_high_dynamics_resampler_32f_xn.h:76).
"""
import numpy as np
import pytest

import oracle
from helpers import TOL_REF, add_code_signal, cn0_to_amplitude, golden_e1_l5_codes, protokernel_distances
from test_tracking_gpu import _bank, _check

pytestmark = pytest.mark.gpu

E1_CHIP_RATE = 1.023e6
L5_CHIP_RATE = 10.23e6
F_L1 = 1575.42e6
F_L5 = 1176.45e6
VEML_SHIFTS = [-1.0, -0.3, 0.0, 0.3, 1.0]  # {-0.5, -0.15, 0, 0.15, 0.5} chips x 2 samples per chip


def _nco(fs, fd, f_carrier, chip_rate, spc, rng):
    """NCO parameters as do_correlation_step passes them (trk.cc:1237-1243): code terms x samples-per-chip"""
    return dict(rem_carr_phase_rad=float(np.float32(rng.uniform(0, 2 * np.pi))),
                phase_step_rad=float(np.float32(2 * np.pi * fd / fs)),
                rem_code_phase_chips=float(np.float32(rng.uniform(0, 1) * spc)),
                code_phase_step_chips=float(np.float32(chip_rate * (1 + fd / f_carrier) / fs * spc)))


def _hold_to_protokernels(tag, out, jobs, codes, x):
    """north_star's 1e-5 against the reference's volk path, where it can be held (helpers.TOL_DISPATCH): on these long Galileo E1 / 50 Msps windows the reference's
    AVX resampler selects other chips than its generic one and the two protokernels sit 1e-3 .. 1e-2 apart -- the GPU (the generic kernel's chips, bit for bit) is
    held to the generic kernel at TOL_REF and to "no further from _u_avx than _generic is"."""
    w_gen, w_avx, w_between = protokernel_distances(out, jobs, codes, x)
    print(f"{tag} on signal taps: |gpu-generic|/|generic| = {w_gen:.3e}" + ("" if w_avx is None else f", |gpu-u_avx|/|u_avx| = {w_avx:.3e}, |u_avx-generic|/|generic| = {w_between:.3e}"))
    assert w_gen <= TOL_REF, (tag, w_gen)
    if w_avx is not None:
        assert w_avx <= max(1e-5, w_between + TOL_REF), (tag, w_avx, w_between)


def test_config4_galileo_e1_50_channels(gpu):
    fs, n, epochs = 32e6, 128000, 2
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(0x5EED0004)
    n_stream = (epochs + 1) * n + 64
    x = (rng.standard_normal(n_stream) + 1j * rng.standard_normal(n_stream)).astype(np.complex64)
    amp = cn0_to_amplitude(45.0, fs)
    sig = []
    for ch in range(6):  # six channels carry a signal: (E1B - E1C)/sqrt(2), no data bit edge inside the windows
        fd = rng.uniform(-4000, 4000)
        ph = rng.uniform(0, 8184)
        rate = E1_CHIP_RATE * (1 + fd / F_L1) / fs * 2.0
        add_code_signal(x, (g["e1b"][ch] - g["e1c"][ch]) / np.sqrt(2.0), fs, rate, ph, fd, amp)
        sig.append((fd, ph, rate))
    codes = [g["e1c"][ch] for ch in range(50)] + [g["e1b"][ch] for ch in range(50)]  # slots 0..49 pilot, 50..99 data
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    jobs = []
    for e in range(epochs):
        for ch in range(50):
            if ch < 6:
                fd, ph, rate = sig[ch]
                start = (8184.0 - ph) / rate + e * (8184.0 / rate)  # where the replica index wraps to 0
                off = int(np.ceil(start))
                p = _nco(fs, fd, F_L1, E1_CHIP_RATE, 2, rng)
                p["rem_code_phase_chips"] = float(np.float32(-(off - start) * rate))
                p["rem_carr_phase_rad"] = float(np.float32((2 * np.pi * fd / fs * off) % (2 * np.pi)))
            else:
                off = int(rng.integers(0, n)) + e * n
                p = _nco(fs, rng.uniform(-4000, 4000), F_L1, E1_CHIP_RATE, 2, rng)
            jobs.append(dict(sample_offset=off, n_samples=n, code_slot=ch, shifts_chips=VEML_SHIFTS, **p))
            jobs.append(dict(sample_offset=off, n_samples=n, code_slot=50 + ch, shifts_chips=[0.0], **p))  # data prompt
    out = b.correlate(jobs)
    worst = _check(out, jobs, codes, x, tol_ref=TOL_REF)
    for j in range(0, 12, 2):  # pilot prompt of the aligned channels: |P| ~ A*N/sqrt(2) = 4*sqrt(2N)
        assert abs(out[j, 2]) > 2.5 * np.sqrt(2 * n), (j, out[j])
        assert abs(out[j + 1, 0]) > 2.5 * np.sqrt(2 * n), (j, out[j + 1])
    print(f"config4 worst |gpu-truth|/sum|x| = {worst:.3e} over {len(jobs)} jobs")
    _hold_to_protokernels("config4", out, jobs[:24], codes, x)
    b.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_config4_high_dynamics_long_window(gpu, mode):
    """N = 128 000 > 65 535: (unsigned)(n*n) wraps inside the window exactly as in the reference"""
    fs, n = 32e6, 128000
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(77 + mode)
    x = (rng.standard_normal(n + 200) + 1j * rng.standard_normal(n + 200)).astype(np.complex64)
    codes = [g["e1c"][10], g["e1b"][10]]
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    jobs = []
    for k in range(4):
        p = _nco(fs, rng.uniform(-4000, 4000), F_L1, E1_CHIP_RATE, 2, rng)
        p["code_phase_rate_step_chips"] = float(np.float32(rng.uniform(-1, 1) * 1e-12))
        p["phase_rate_step_rad"] = float(np.float32(rng.uniform(-1, 1) * 2e-11))
        jobs.append(dict(sample_offset=int(rng.integers(0, 190)), n_samples=n, code_slot=k % 2, high_dyn=mode,
                         shifts_chips=VEML_SHIFTS if k % 2 == 0 else [0.0], **p))
    out = b.correlate(jobs)
    # same bars as tests/test_tracking_gpu.py::test_high_dynamics_modes: truth at 1e-6; the reference's own HD rotator
    # never renormalises its second phasor, so it is compared at its QA tolerance only
    from helpers import oracle_job, scale_err
    for j, job in enumerate(jobs):
        o32, t64, sabs = oracle_job(codes[job["code_slot"]], x, job)
        nt = len(job["shifts_chips"])
        assert np.all(scale_err(out[j, :nt], t64, sabs) <= 1e-6), (mode, j, out[j, :nt], t64)
        assert np.all(np.abs(out[j, :nt] - o32) <= 1e-3 * sabs), (mode, j, out[j, :nt], o32)
    b.close()


def test_config5_multi_constellation_256_channels(gpu):
    fs = 50e6
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(0x5EED0005)
    n_l1, n_e1, n_l5 = 50000, 200000, 50000
    len_a = n_e1 + n_l1 + 64   # RF stream A: L1 / E1 band
    len_b = 2 * n_l5 + 64      # RF stream B: L5 band
    x = (rng.standard_normal(len_a + len_b) + 1j * rng.standard_normal(len_a + len_b)).astype(np.complex64)
    amp = cn0_to_amplitude(47.0, fs)
    # one signal per constellation so that some accumulators are signal-dominated
    add_code_signal(x[:len_a], oracle.ca_code(3).astype(np.float32), fs, 1.023e6 / fs, 100.0, 1500.0, amp)
    add_code_signal(x[:len_a], (g["e1b"][4] - g["e1c"][4]) / np.sqrt(2.0), fs, 2 * E1_CHIP_RATE / fs, 500.0, -2200.0, amp)
    add_code_signal(x[len_a:], g["l5q"][6], fs, L5_CHIP_RATE / fs, 2000.0, 900.0, amp)
    codes, jobs = [], []

    def slot(code):
        codes.append(np.ascontiguousarray(code, np.float32))
        return len(codes) - 1

    for ch in range(96):  # GPS L1 C/A, E/P/L
        s = slot(oracle.ca_code(ch % 32 + 1))
        fd = 1500.0 if ch == 2 else rng.uniform(-5000, 5000)
        jobs.append(dict(sample_offset=int(rng.integers(0, len_a - n_l1)), n_samples=n_l1, code_slot=s, shifts_chips=[-0.5, 0.0, 0.5],
                         **_nco(fs, fd, F_L1, 1.023e6, 1, rng)))
    for ch in range(96):  # Galileo E1: VE/E/P/L/VL on the pilot + data prompt
        sp, sd = slot(g["e1c"][ch % 50]), slot(g["e1b"][ch % 50])
        fd = -2200.0 if ch == 4 else rng.uniform(-5000, 5000)
        off = int(rng.integers(0, len_a - n_e1))
        p = _nco(fs, fd, F_L1, E1_CHIP_RATE, 2, rng)
        jobs.append(dict(sample_offset=off, n_samples=n_e1, code_slot=sp, shifts_chips=VEML_SHIFTS, **p))
        jobs.append(dict(sample_offset=off, n_samples=n_e1, code_slot=sd, shifts_chips=[0.0], **p))
    for ch in range(64):  # GPS L5 (pilot Q replica), E/P/L, second RF stream
        s = slot(g["l5q"][ch % 32])
        fd = rng.uniform(-4000, 4000)
        jobs.append(dict(sample_offset=len_a + int(rng.integers(0, len_b - n_l5)), n_samples=n_l5, code_slot=s, shifts_chips=[-0.5, 0.0, 0.5],
                         **_nco(fs, fd, F_L5, L5_CHIP_RATE, 1, rng)))
    assert len(jobs) == 96 + 2 * 96 + 64
    b = _bank(gpu, codes, max_len=10230)
    b.set_stream_host(x)
    out = b.correlate(jobs)
    worst = _check(out, jobs, codes, x, tol_ref=TOL_REF)
    print(f"config5 worst |gpu-truth|/sum|x| = {worst:.3e} over {len(jobs)} jobs (256 channels)")
    sig = [j for j, job in enumerate(jobs) if (job["code_slot"] == jobs[2]["code_slot"]) or (job is jobs[96 + 2 * 4]) or (job is jobs[96 + 2 * 4 + 1])]
    _hold_to_protokernels("config5", out[sig], [jobs[j] for j in sig], codes, x)
    # order independence: the same jobs reversed give the same numbers job by job
    out_r = b.correlate(jobs[::-1])
    assert np.array_equal(out_r[::-1].view(np.float32), out.view(np.float32))
    b.close()


@pytest.mark.parametrize("splits", [0, 1, 3, 4, 8, 16])
def test_windowed_code_table_chip_selection_exact(gpu, splits):
    """Long codes with split windows stage only the code samples a segment can touch (multicorrelator.hip, bank_window_floats).
    With x[n] = 1, zero carrier and integer-valued sums every float32 addition is exact, so the result equals sum_n code[k_t[n]]
    exactly iff every chip index -- including the ones at segment edges, in the masked head / tail chunks and across the code wrap --
    matches the oracle's, whatever the split (splits = 1 keeps the whole code in LDS: the two stagings must agree)."""
    g = golden_e1_l5_codes()
    codes = [g["e1c"][3], g["l5q"][7], g["e1b"][9]]
    n_stream = 260000
    x = np.ones(n_stream, np.complex64)
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    b.set_splits(splits)
    rng = np.random.default_rng(40 + splits)
    jobs = []
    for i in range(24):
        slot = i % 3
        L = len(codes[slot])
        n = int(rng.choice([50000, 128000, 200000, 31337]))
        step = np.float32((L / n) * rng.uniform(0.98, 1.02))          # about one code period per window
        rem = np.float32(rng.uniform(0.0, 2.0))
        shifts = [-1.0, -0.3, 0.0, 0.3, 1.0] if slot == 0 else ([0.0] if slot == 2 else [-0.5, 0.0, 0.5])
        jobs.append(dict(sample_offset=int(rng.integers(0, n_stream - n)), n_samples=n, code_slot=slot, shifts_chips=shifts, rem_carr_phase_rad=0.0,
                         phase_step_rad=0.0, rem_code_phase_chips=float(rem), code_phase_step_chips=float(step)))
    # a window that runs over more than one code period (wraps inside a segment) and one with a large negative starting index
    jobs.append(dict(sample_offset=11, n_samples=100000, code_slot=1, shifts_chips=[-0.5, 0.0, 0.5], rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                     rem_code_phase_chips=0.25, code_phase_step_chips=0.25))
    jobs.append(dict(sample_offset=7, n_samples=60000, code_slot=0, shifts_chips=[-1.0, -0.3, 0.0, 0.3, 1.0], rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                     rem_code_phase_chips=5000.5, code_phase_step_chips=0.064))
    out = b.correlate(jobs)
    for j, job in enumerate(jobs):
        sh = np.asarray(job["shifts_chips"], np.float32)
        L = len(codes[job["code_slot"]])
        idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, L, False)
        expect = np.array([codes[job["code_slot"]][idx[t]].astype(np.float64).sum() for t in range(len(sh))])
        got = out[j, :len(sh)]
        assert np.array_equal(got.real.astype(np.float64), expect), (splits, j, job, got, expect)
        assert np.all(got.imag == 0)
    b.close()


@pytest.mark.parametrize("splits", [0, 1, 4])
def test_pair_fusion_is_bit_identical(gpu, splits):
    """track_pilot runs a second, single-tap correlator with the data code over the window the pilot's VE/E/P/L/VL just used
    (trk.cc:1246-1256).  The bank computes such a job inside the job in front of it (gsh_bank_set_pair_fusion); every tap remains the
    same sum of the same products in the same order, so the outputs with and without fusion are equal to the bit -- Galileo E1
    (5 + 1 taps, windowed long code), GPS L5 (3 + 1), C/A-length codes (whole tables, two of them in LDS), odd offsets, short windows,
    and a lone single-tap job that has nothing to ride on."""
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(300 + splits)
    codes = [g["e1c"][1], g["e1b"][1], g["l5q"][2], g["l5i"][2], oracle.ca_code(4), oracle.ca_code(9)]
    n_stream = 420000
    x = (rng.standard_normal(n_stream) + 1j * rng.standard_normal(n_stream)).astype(np.complex64)
    jobs = []

    def pair(n, pilot, data, shifts, chip_rate, spc, fs, data_shift=0.0):
        off = int(rng.integers(0, n_stream - n))
        p = _nco(fs, rng.uniform(-4000, 4000), F_L1, chip_rate, spc, rng)
        jobs.append(dict(sample_offset=off, n_samples=n, code_slot=pilot, shifts_chips=shifts, **p))
        jobs.append(dict(sample_offset=off, n_samples=n, code_slot=data, shifts_chips=[data_shift], **p))

    for _ in range(3):
        pair(128000, 0, 1, VEML_SHIFTS, E1_CHIP_RATE, 2, 32e6)
        pair(50000, 2, 3, [-0.5, 0.0, 0.5], L5_CHIP_RATE, 1, 50e6)
        pair(25000, 4, 5, [-0.5, 0.0, 0.5], 1.023e6, 1, 25e6)
    pair(30001, 0, 1, VEML_SHIFTS, E1_CHIP_RATE, 2, 32e6, data_shift=0.3)   # a fused tap away from the prompt
    pair(777, 4, 5, [-0.5, 0.0, 0.5], 1.023e6, 1, 25e6)                       # shorter than one chunk: masked paths only
    jobs.append(dict(sample_offset=5, n_samples=40000, code_slot=3, shifts_chips=[0.0], **_nco(50e6, 100.0, F_L5, L5_CHIP_RATE, 1, rng)))  # no leader
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    b.set_splits(splits)
    b.set_pair_fusion(False)
    plain = b.correlate(jobs).copy()
    b.set_pair_fusion(True)
    fused = b.correlate(jobs)
    assert np.array_equal(plain.view(np.uint32), fused.view(np.uint32))
    worst = _check(fused, jobs, codes, x, tol_ref=TOL_REF)
    assert worst < 1e-6
    b.close()
