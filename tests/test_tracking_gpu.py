"""GPU parity tests of the multicorrelator (through the C ABI) against the oracle.

Modelled on the reference's own correlator test
(tests/unit-tests/signal-processing-blocks/tracking/cpu_multicorrelator_real_codes_test.cc:65-180), which only
asserts EXPECT_NO_THROW; here every call is also compared with oracle/ (float32 restatement pinned to the
reference + float64 truth).

Bars (tests/helpers.py): chip selection bit-exact (checked through exact equality on a noise-free,
carrier-free input); |gpu - truth| / sum|x| <= 1e-6 (north_star's 1e-5, with margin);
|gpu - generic| / |generic| <= 5e-5 on taps that hold a signal.
"""
import os
import sys

import numpy as np
import pytest

import oracle
from helpers import (TOL_DISPATCH, TOL_REF, oracle_job, scale_err, synth_gps_l1_stream, tracking_params_for)

pytestmark = pytest.mark.gpu

TOL_SCALE_GPU = 1e-6


def _bank(gpu, codes, max_len=None):
    from gnss_sdr_amd.tracking import CorrelatorBank
    max_len = max_len or max(len(c) for c in codes)
    b = CorrelatorBank(len(codes), max_len, device=gpu)
    for i, c in enumerate(codes):
        b.set_code(i, c)
    return b


def _check(out, jobs, codes, x, tol_scale=TOL_SCALE_GPU, tol_ref=None):
    worst = 0.0
    for j, job in enumerate(jobs):
        code = codes[job.get("code_slot", 0)]
        o32, t64, sabs = oracle_job(code, x, job)
        nt = len(job["shifts_chips"])
        g = out[j, :nt]
        e = scale_err(g, t64, sabs)
        worst = max(worst, float(e.max()))
        assert np.all(e <= tol_scale), f"job {j}: |gpu-truth|/sum|x| = {e} > {tol_scale}; gpu={g} truth={t64}"
        assert np.all(out[j, nt:] == 0), f"job {j}: taps beyond n_taps must be zero"
        # at least as accurate as the reference's own float32 kernel (plus float32 output rounding)
        eo = scale_err(o32, t64, sabs)
        assert np.all(e <= eo + 2e-7), f"job {j}: gpu error {e} worse than reference _generic {eo}"
        if tol_ref is not None:
            rel = np.abs(g - o32) / np.abs(o32)
            strong = np.abs(t64) > 0.01 * sabs
            assert np.all(rel[strong] <= tol_ref), f"job {j}: |gpu-generic|/|generic| = {rel} > {tol_ref}"
    return worst


def test_reference_unit_test_case(gpu):
    """The exact parameter set of CpuMulticorrelatorRealCodesTest.MeasureExecutionTime (:113-133): uniform [0,1)
    input, 3 taps at -0.5/0/+0.5, phase step 0.1 rad, code step 0.3, code rate 1e-5, rem 0.4, sizes 2048/4096/8192,
    driven through the Cpu_Multicorrelator_Real_Codes-shaped object with its default high-dynamics resampler
    and the 6-argument call the reference test uses."""
    from gnss_sdr_amd.tracking import HipMulticorrelatorRealCodes
    rng = np.random.default_rng(7)
    vlen = 8192
    in_cpu = (rng.random(2 * vlen) + 1j * rng.random(2 * vlen)).astype(np.complex64)
    ca = oracle.ca_code(1)
    outs = np.zeros(3, np.complex64)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    mc = HipMulticorrelatorRealCodes(gpu)
    assert mc.init(vlen, 3)
    assert mc.set_input_output_vectors(outs, in_cpu)
    assert mc.set_local_code_and_taps(1023, ca, shifts)
    for n in (2048, 4096, 8192):
        assert mc.Carrier_wipeoff_multicorrelator_resampler(0.0, 0.1, 0.4, 0.3, 0.00001, n)
        job = dict(n_samples=n, shifts_chips=shifts, rem_carr_phase_rad=0.0, phase_step_rad=0.1,
                   rem_code_phase_chips=0.4, code_phase_step_chips=0.3, code_phase_rate_step_chips=0.00001, high_dyn=2)
        o32, t64, sabs = oracle_job(ca, in_cpu, job)
        assert np.all(scale_err(outs, t64, sabs) <= TOL_SCALE_GPU), (n, outs, t64)
        strong = np.abs(t64) > 0.01 * sabs
        assert np.all((np.abs(outs - o32) / np.abs(o32))[strong] <= TOL_REF), (n, outs, o32)
    # standard resampler, 7-argument form, taps mutated in place between calls (trk.cc:2132-2146)
    mc.set_high_dynamics_resampler(False)
    shifts[0], shifts[2] = -0.15, 0.15
    assert mc.Carrier_wipeoff_multicorrelator_resampler(1.0, 0.05, 0.0, 0.2, 0.25, 0.0, 4096)
    job = dict(n_samples=4096, shifts_chips=shifts, rem_carr_phase_rad=1.0, phase_step_rad=0.05,
               rem_code_phase_chips=0.2, code_phase_step_chips=0.25)
    o32, t64, sabs = oracle_job(ca, in_cpu, job)
    assert np.all(scale_err(outs, t64, sabs) <= TOL_SCALE_GPU)
    assert mc.free()
    mc.close()


def test_chip_selection_bit_exact(gpu):
    """With x[n] = 1, zero carrier and integer-valued sums every float32 addition is exact, so the GPU result
    equals sum_n code[k_t[n]] exactly iff every chip index matches the oracle's (itself bit-exact with the
    reference's resampler, tests/test_oracle_golden.py::test_oracle_equals_live_reference)."""
    fs = 25e6
    n = 25000
    x = np.ones(2 * n + 7, np.complex64)
    codes = [oracle.ca_code(p) for p in (1, 5, 17, 32)]
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    rng = np.random.default_rng(11)
    jobs = []
    for i in range(64):
        fd = rng.uniform(-5000, 5000)
        p = tracking_params_for(fs, fd, rng)
        jobs.append(dict(sample_offset=int(rng.integers(0, n)), n_samples=n, code_slot=i % 4,
                         shifts_chips=[-0.5, 0.0, 0.5], rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                         rem_code_phase_chips=p["rem_code_phase_chips"], code_phase_step_chips=p["code_phase_step_chips"]))
    # extra shapes: 5 taps, one tap, multi-period windows (wrap path), tiny window, odd offsets
    jobs.append(dict(sample_offset=1, n_samples=8111, code_slot=0, shifts_chips=[-0.5, -0.25, 0.0, 0.25, 0.5],
                     rem_code_phase_chips=0.4, code_phase_step_chips=0.3))
    jobs.append(dict(sample_offset=3, n_samples=4001, code_slot=1, shifts_chips=[0.0], rem_code_phase_chips=0.9,
                     code_phase_step_chips=0.25575))
    jobs.append(dict(sample_offset=0, n_samples=1, code_slot=2, shifts_chips=[-0.5, 0.0, 0.5], rem_code_phase_chips=0.1,
                     code_phase_step_chips=0.04))
    jobs.append(dict(sample_offset=5, n_samples=513, code_slot=3, shifts_chips=[-1.5, 0.0, 1.5], rem_code_phase_chips=0.0,
                     code_phase_step_chips=1.7))
    for group in (jobs[:64], jobs[64:65], jobs[65:66], jobs[66:]):
        out = b.correlate(group)
        for j, job in enumerate(group):
            sh = np.asarray(job["shifts_chips"], np.float32)
            idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, 1023, False)
            expect = np.array([codes[job["code_slot"]][idx[t]].astype(np.float64).sum() for t in range(len(sh))])
            got = out[j, :len(sh)]
            assert np.array_equal(got.real.astype(np.float64), expect), (job, got, expect)
            assert np.all(got.imag == 0)
    b.close()


def _derived_tap_jobs(rng, n, n_codes):
    """E/P/L jobs aimed at the derived-tap trips (csrc/mcorr_device.h derived_lookup): shift sets that qualify ((-s, 0, 1 - s), s a multiple of 2^-8) and
    sets that do not, code steps that put the chip phase exactly ON the compare threshold (dyadic steps: ties), code phases below zero and beyond one,
    windows whose chip range crosses every power of two up to the code length."""
    shift_sets = [[-0.5, 0.0, 0.5], [-0.25, 0.0, 0.75], [-1.0, 0.0, 0.0], [-0.75, 0.0, 0.25], [-0.00390625, 0.0, 0.99609375], [-0.3, 0.0, 0.3], [-0.5, 0.0, 0.75]]
    steps = [1.023e6 / 25e6, 1.023e6 / 4e6, 0.5, 0.25, 0.125, 0.0625, 0.03125, 1.0, 0.040919998, 0.0409200004]
    rems = [0.0, 0.5, 0.25, 0.999, -0.3, 1.7, 0.4999999, 0.5000001, 1.0 / 3.0]
    jobs = []
    for sh in shift_sets:
        for step in steps:
            rem = float(np.float32(rems[len(jobs) % len(rems)]))
            step32 = float(np.float32(step))
            length = int(min(n, (1023 + 20) / step32))  # stay inside one code period (+ margin): the path without the per-sample wrap
            jobs.append(dict(sample_offset=int(rng.integers(0, 64)), n_samples=length, code_slot=len(jobs) % n_codes, shifts_chips=sh, rem_carr_phase_rad=0.0,
                             phase_step_rad=0.0, rem_code_phase_chips=rem, code_phase_step_chips=step32))
    return jobs


def test_derived_taps_select_the_reference_chips(gpu):
    """Integer-valued samples, zero carrier: every float32 sum is exact, so a tap's output equals sum_n x[n] code[k_t[n]] iff every chip index is the
    oracle's.  The weights make a swapped pair of indices visible (with x = 1 only the count of each code value would be)."""
    n = 26000
    rng = np.random.default_rng(2024)
    xr = rng.integers(-7, 8, 2 * n + 128).astype(np.float32)
    x = xr.astype(np.complex64)
    codes = [oracle.ca_code(p) for p in (2, 9, 23)]
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    jobs = _derived_tap_jobs(rng, n, len(codes))
    for k in range(0, len(jobs), 10):
        group = jobs[k:k + 10]
        out = b.correlate(group)
        for j, job in enumerate(group):
            sh = np.asarray(job["shifts_chips"], np.float32)
            idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, 1023, False)
            seg = xr[job["sample_offset"]:job["sample_offset"] + job["n_samples"]].astype(np.float64)
            expect = np.array([(codes[job["code_slot"]][idx[t]].astype(np.float64) * seg).sum() for t in range(3)])
            got = out[j, :3]
            assert np.array_equal(got.real.astype(np.float64), expect), (job, got, expect)
            assert np.all(got.imag == 0)
    b.close()


def test_paired_taps_random_sweep(gpu):
    """600 random E/P/L jobs with a one-chip early-late spacing -- code steps from 0.01 to 2 chips per sample (a third of them dyadic multiples of 2^-10, where
    chip phases land exactly on tap boundaries), code phases in [-1, 2), window starts anywhere, lengths from one sample to a code period -- against the oracle's chips."""
    rng = np.random.default_rng(31337)
    n_max = 30000
    xr = rng.integers(-7, 8, 2 * n_max + 128).astype(np.float32)
    codes = [oracle.ca_code(p) for p in (3, 11, 29)]
    b = _bank(gpu, codes)
    b.set_stream_host(xr.astype(np.complex64))
    jobs = []
    for i in range(600):
        kind = i % 3
        if kind == 0:
            step = float(np.float32(rng.uniform(0.01, 2.0)))
        elif kind == 1:
            step = float(np.float32(rng.integers(8, 2048) / 1024.0))
        else:
            step = float(np.float32(1.023e6 / rng.choice([2e6, 4e6, 5e6, 10e6, 12.5e6, 16.368e6, 25e6, 50e6])))
        s = float(np.float32(rng.choice([0.5, 0.25, 0.75, 1.0, 0.125, 0.3])))
        sh = [float(np.float32(-s)), 0.0, float(np.float32(np.float32(-s) + np.float32(1.0)))]
        length = int(min(n_max, max(1, rng.integers(1, int((1023 + 20) / step) + 1))))
        jobs.append(dict(sample_offset=int(rng.integers(0, n_max)), n_samples=length, code_slot=i % 3, shifts_chips=sh, rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                         rem_code_phase_chips=float(np.float32(rng.uniform(-1.0, 2.0))), code_phase_step_chips=step))
    for k in range(0, len(jobs), 50):
        group = jobs[k:k + 50]
        out = b.correlate(group)
        for j, job in enumerate(group):
            sh = np.asarray(job["shifts_chips"], np.float32)
            idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, 1023, False)
            seg = xr[job["sample_offset"]:job["sample_offset"] + job["n_samples"]].astype(np.float64)
            expect = np.array([(codes[job["code_slot"]][idx[t]].astype(np.float64) * seg).sum() for t in range(3)])
            assert np.array_equal(out[j, :3].real.astype(np.float64), expect), (job, out[j, :3], expect)
            assert np.all(out[j, :3].imag == 0)
    b.close()


def test_two_wave_work_groups_of_large_launches(gpu):
    """A launch of at least 5 120 E/P/L jobs runs the 128-thread kernels (two waves per job, csrc/multicorrelator_t128.hip), smaller ones the 256-thread kernels
    (mcorr_launch picks).  The same 5 400 jobs in one launch and in batches of 900: chip selection exact in both (integer-valued carrier-free input: the sums are
    exact in float32 whatever the order of summation, so the two launches must agree bit for bit and with the oracle's chips), and with noise and a carrier both
    within the accumulator bar of the float64 truth."""
    rng = np.random.default_rng(424242)
    n_max = 9000
    xr = rng.integers(-7, 8, 2 * n_max + 128).astype(np.float32)
    codes = [oracle.ca_code(p) for p in (5, 17, 23, 31)]
    b = _bank(gpu, codes)
    b.set_stream_host(xr.astype(np.complex64))
    jobs = []
    for i in range(5400):
        step = float(np.float32(1.023e6 / rng.choice([4e6, 5e6, 10e6, 12.5e6, 25e6]))) if i % 2 else float(np.float32(rng.uniform(0.03, 0.5)))
        length = int(min(n_max, max(1, rng.integers(1, int(1040 / step) + 1))))
        jobs.append(dict(sample_offset=int(rng.integers(0, n_max)), n_samples=length, code_slot=i % 4, shifts_chips=[-0.5, 0.0, 0.5], rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                         rem_code_phase_chips=float(np.float32(rng.uniform(-0.5, 1.5))), code_phase_step_chips=step))
    big = b.correlate(jobs)
    small = np.concatenate([b.correlate(jobs[k:k + 900]) for k in range(0, len(jobs), 900)], axis=0)
    assert np.array_equal(big.view(np.uint32), small.view(np.uint32))
    for j in range(0, len(jobs), 9):
        job = jobs[j]
        idx = oracle.code_indices(job["n_samples"], np.asarray(job["shifts_chips"], np.float32), job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, 1023, False)
        seg = xr[job["sample_offset"]:job["sample_offset"] + job["n_samples"]].astype(np.float64)
        expect = np.array([(codes[job["code_slot"]][idx[t]].astype(np.float64) * seg).sum() for t in range(3)])
        assert np.array_equal(big[j, :3].real.astype(np.float64), expect), (job, big[j, :3], expect)
    # noise + carrier: both work-group sizes against the float64 truth
    x = (rng.standard_normal(2 * n_max + 128) + 1j * rng.standard_normal(2 * n_max + 128)).astype(np.complex64)
    b.set_stream_host(x)
    for job in jobs:
        job.update(rem_carr_phase_rad=float(np.float32(rng.uniform(0, 6.28))), phase_step_rad=float(np.float32(rng.uniform(-0.02, 0.02))))
    big = b.correlate(jobs)
    small = np.concatenate([b.correlate(jobs[k:k + 900]) for k in range(0, len(jobs), 900)], axis=0)
    pick = list(range(0, len(jobs), 27))
    w_big = _check(big[pick], [jobs[j] for j in pick], codes, x)
    w_small = _check(small[pick], [jobs[j] for j in pick], codes, x)
    print(f"two-wave kernels: worst |gpu-truth|/sum|x| = {w_big:.2e}; four-wave kernels: {w_small:.2e}")
    b.close()


def test_derived_taps_on_long_codes(gpu):
    """The same on 10 230-chip codes (chip indices up to 2^13.3: the binades 1024 .. 8192), whole windows and the windowed code table with automatic splits."""
    fs, n = 25e6, 25000
    rng = np.random.default_rng(99)
    codes = [(2 * rng.integers(0, 2, 10230) - 1).astype(np.int32) for _ in range(2)]
    xr = rng.integers(-7, 8, 2 * n + 128).astype(np.float32)
    b = _bank(gpu, codes)
    b.set_stream_host(xr.astype(np.complex64))
    step = float(np.float32(10.23e6 / fs))
    jobs = []
    for i in range(24):
        sh = [[-0.5, 0.0, 0.5], [-0.25, 0.0, 0.75], [-1.0, 0.0, 0.0]][i % 3]
        jobs.append(dict(sample_offset=int(rng.integers(0, n)), n_samples=n - int(rng.integers(0, 3)), code_slot=i % 2, shifts_chips=sh, rem_carr_phase_rad=0.0,
                         phase_step_rad=0.0, rem_code_phase_chips=float(np.float32(rng.uniform(-0.5, 1.5))), code_phase_step_chips=step))
    out = b.correlate(jobs)
    for j, job in enumerate(jobs):
        sh = np.asarray(job["shifts_chips"], np.float32)
        idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], step, 0.0, 10230, False)
        seg = xr[job["sample_offset"]:job["sample_offset"] + job["n_samples"]].astype(np.float64)
        expect = np.array([(codes[job["code_slot"]][idx[t]].astype(np.float64) * seg).sum() for t in range(3)])
        assert np.array_equal(out[j, :3].real.astype(np.float64), expect), (job, out[j, :3], expect)
    b.close()


def test_paired_taps_near_2048_chips_and_with_tiny_code_phases(gpu):
    """The paired trips take two floors per instruction (csrc/mcorr_device.h, GSH_MC_PKRTZ): the chip-index chains run on constants scaled by 2^-24 and
    v_cvt_pkrtz_f16_f32 leaves floor(u) as the half-precision bit pattern -- exact for 1 <= u < 2048 and as long as no scaled constant is denormal.  Here: 2 046-chip
    codes (BeiDou B1I's length: the last chips lie beyond the bound, so the waves that hold them must take the one-floor form), code phases of 1e-37, 1e-30, -1e-33
    (their scaled values would be denormal: no trip may take the scaled chains), exact zeros, and steps that put u on integers.  Integer-valued samples and no carrier:
    a tap's sum is exact, so it equals the oracle's iff every chip index is the reference's."""
    rng = np.random.default_rng(2048)
    n_max = 52000
    xr = rng.integers(-7, 8, 2 * n_max + 128).astype(np.float32)
    codes = [(2 * rng.integers(0, 2, 2046) - 1).astype(np.int32) for _ in range(3)]
    b = _bank(gpu, codes)
    b.set_stream_host(xr.astype(np.complex64))
    rems = [0.0, 1e-37, 1e-30, -1e-33, 0.25, 0.999, -0.3, 1.7, 1.0 / 3.0]
    steps = [2.046e6 / 25e6, 2.046e6 / 50e6, 0.0625, 0.5, 0.040919998, 2.046e6 / 10e6]
    jobs = []
    for i, (rem, step) in enumerate((r, s) for r in rems for s in steps):
        step32 = float(np.float32(step))
        length = int(min(n_max, (2046 + 20) / step32))
        s_ = [0.5, 0.25, 1.0][i % 3]
        jobs.append(dict(sample_offset=int(rng.integers(0, 64)), n_samples=length, code_slot=i % 3, shifts_chips=[-s_, 0.0, 1.0 - s_], rem_carr_phase_rad=0.0,
                         phase_step_rad=0.0, rem_code_phase_chips=float(np.float32(rem)), code_phase_step_chips=step32))
    for k in range(0, len(jobs), 9):
        group = jobs[k:k + 9]
        out = b.correlate(group)
        for j, job in enumerate(group):
            sh = np.asarray(job["shifts_chips"], np.float32)
            idx = oracle.code_indices(job["n_samples"], sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, 2046, False)
            seg = xr[job["sample_offset"]:job["sample_offset"] + job["n_samples"]].astype(np.float64)
            expect = np.array([(codes[job["code_slot"]][idx[t]].astype(np.float64) * seg).sum() for t in range(3)])
            assert np.array_equal(out[j, :3].real.astype(np.float64), expect), (job, out[j, :3], expect)
            assert np.all(out[j, :3].imag == 0)
    # the same jobs in one launch of the two-wave kernels (>= 5 120 jobs): bit-identical sums
    many = (jobs * (5120 // len(jobs) + 1))[:5200]
    big = b.correlate(many)
    ref = np.concatenate([b.correlate(jobs[k:k + 9]) for k in range(0, len(jobs), 9)], axis=0)
    assert np.array_equal(big[:len(jobs)].view(np.uint32), ref.view(np.uint32)) and np.array_equal(big[len(jobs):2 * len(jobs)].view(np.uint32), ref.view(np.uint32))
    b.close()


_DERIVED_AB_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import oracle
from test_tracking_gpu import _bank, _derived_tap_jobs
from helpers import tracking_params_for
rng = np.random.default_rng(5)
n = 26000
x = (rng.standard_normal(2 * n + 128) + 1j * rng.standard_normal(2 * n + 128)).astype(np.complex64)
codes = [oracle.ca_code(p) for p in (2, 9, 23)]
b = _bank(0, codes)
b.set_stream_host(x)
jobs = _derived_tap_jobs(rng, n, 3)
for job in jobs:
    p = tracking_params_for(25e6, float(rng.uniform(-5000, 5000)), rng)
    job["rem_carr_phase_rad"] = p["rem_carr_phase_rad"]
    job["phase_step_rad"] = p["phase_step_rad"]
outs = [b.correlate(jobs[k:k + 10]) for k in range(0, len(jobs), 10)]
np.save(sys.argv[1], np.concatenate(outs))
"""


def test_paired_taps_are_bit_identical(gpu, tmp_path):
    """Random complex samples with a carrier, the same jobs with the paired-tap trips on (default) and off (GSH_MC_PACKED_BODY=3, read once per process):
    same chips, same products, same order of summation -- the outputs must agree bit for bit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _DERIVED_AB_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    outs = {}
    for body in ("1", "3"):
        f = str(tmp_path / f"out_{body}.npy")
        r = subprocess.run([sys.executable, "-c", script, f], cwd=root, env=dict(os.environ, GSH_MC_PACKED_BODY=body), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[body] = np.load(f)
    assert outs["1"].shape == outs["3"].shape and outs["1"].shape[0] >= 60
    assert np.array_equal(outs["1"].view(np.uint32), outs["3"].view(np.uint32))


def test_even_tap_counts_and_one_sided_shifts_on_a_windowed_code(gpu):
    """Long codes (GPS L5: 10 230 chips) are staged as a per-segment window of the code whose length the host derives from the jobs'
    REAL taps; jobs whose tap count is not 1 / 3 / 5 run in the next wider kernel with padded tap slots.  Tap sets sitting far to one side
    of zero ({20, 21}, {-40 .. -38.5}, a lone tap at +30 in a batch of E/P/L jobs) must select exactly the oracle's chips -- a padded
    slot at shift 0 would widen the device's window beyond what the host sized and turn a valid job into NaN."""
    fs, n = 25e6, 25000
    rng = np.random.default_rng(77)
    codes = [(2 * rng.integers(0, 2, 10230) - 1).astype(np.int32) for _ in range(2)]
    x = np.ones(2 * n + 7, np.complex64)
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    step = float(np.float32(10.23e6 / fs))
    tapsets = [[20.0, 21.0], [-40.0, -39.5, -39.0, -38.5], [16.5, 17.0, 17.5, 18.0, 18.5, 19.0], [-30.0, -29.0, -28.0, -27.0, -26.0, -25.0, -24.0], [30.0], [-0.5, 0.0, 0.5]]
    groups = [[ts] * 8 for ts in tapsets[:4]]                 # uniform batches: 2, 4, 6, 7 taps
    groups.append([tapsets[4], tapsets[5]] * 6)               # a lone far tap among E/P/L jobs
    groups.append([tapsets[0], tapsets[5], tapsets[1]] * 4)   # mixed
    for group in groups:
        jobs = [dict(sample_offset=int(rng.integers(0, n)), n_samples=n, code_slot=i % 2, shifts_chips=ts, rem_carr_phase_rad=0.0, phase_step_rad=0.0,
                     rem_code_phase_chips=float(np.float32(rng.uniform(0, 1))), code_phase_step_chips=step) for i, ts in enumerate(group)]
        out = b.correlate(jobs)
        for j, job in enumerate(jobs):
            sh = np.asarray(job["shifts_chips"], np.float32)
            idx = oracle.code_indices(n, sh, job["rem_code_phase_chips"], step, 0.0, 10230, False)
            expect = np.array([codes[job["code_slot"]][idx[t]].astype(np.float64).sum() for t in range(len(sh))])
            got = out[j, :len(sh)]
            assert np.all(np.isfinite(got.real)), (job, got)
            assert np.array_equal(got.real.astype(np.float64), expect), (job, got, expect)
            assert np.all(got.imag == 0)
    b.close()


def test_config2_tracking_parity(gpu):
    """BASELINE config 2 shape: GPS L1 C/A, fs = 25 Msps, N = 25 000, 3-tap E/P/L, 32 channels (PRN 1..32) reading
    windows of one shared stream with 8 embedded signals at 45 dB-Hz, a few epochs each."""
    fs = 25e6
    n = 25000
    epochs = 3
    rng = np.random.default_rng(0x5EED0003)
    sig_prns = list(range(1, 9))
    dop = rng.uniform(-5000, 5000, 8)
    cph = rng.uniform(0, 1023, 8)
    x = synth_gps_l1_stream((epochs + 2) * n, fs, sig_prns, dop, cph)
    codes = [oracle.ca_code(p) for p in range(1, 33)]
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    jobs = []
    for e in range(epochs):
        for ch in range(32):
            if ch < 8:
                # aligned to the embedded signal: window starts where the code phase wraps to ~0
                fd = dop[ch]
                f_code = 1.023e6 * (1 + fd / 1575.42e6)
                start = (1023.0 - cph[ch]) / f_code * fs + e * (1023.0 / f_code * fs)
                off = int(np.ceil(start))
                rem_code = (off - start) * f_code / fs
                p = tracking_params_for(fs, fd, rng)
                p["rem_code_phase_chips"] = float(np.float32(-rem_code))  # code already advanced by rem chips
                p["rem_carr_phase_rad"] = float(np.float32((2 * np.pi * fd / fs * off) % (2 * np.pi)))
            else:
                off = int(rng.integers(0, n)) + e * n
                p = tracking_params_for(fs, rng.uniform(-5000, 5000), rng)
            jobs.append(dict(sample_offset=off, n_samples=n, code_slot=ch, shifts_chips=[-0.5, 0.0, 0.5], **p))
    out = b.correlate(jobs)
    worst = _check(out, jobs, codes, x, tol_ref=TOL_REF)
    # the aligned channels must actually see their signal: |P| ~ A*N = 1257 against a noise floor of sqrt(2N) = 224
    for j in range(8):
        assert abs(out[j, 1]) > 3 * np.sqrt(2 * n), (j, out[j])
    print(f"config2 worst |gpu-truth|/sum|x| = {worst:.3e}")
    # ---- information SURVEY.md section 7 asks for, and the numbers TOL_REF is derived from: how far the GPU sits from the reference's two CPU
    # protokernels on the taps that hold a signal, how far those sit from each other, and how many samples the AVX resampler puts on another chip
    # than the generic one (the GPU selects the generic kernel's chips bit for bit: test_chip_selection_bit_exact)
    w_gen = w_avx = w_between = 0.0
    flips = flip_samples = 0
    R = oracle.ref()
    for j, job in enumerate(jobs):
        code = codes[job["code_slot"]]
        o32, t64, sabs = oracle_job(code, x, job)
        strong = np.abs(t64) > 0.01 * sabs
        if not np.any(strong):
            continue
        g = out[j, :3]
        w_gen = max(w_gen, float((np.abs(g - o32) / np.abs(o32))[strong].max()))
        if R is not None and R.ref_simd_supported():
            win = x[job["sample_offset"]:job["sample_offset"] + n]
            avx = oracle.ref_mcorr(code, job["shifts_chips"], win, job["rem_carr_phase_rad"], job["phase_step_rad"], job["rem_code_phase_chips"],
                                   job["code_phase_step_chips"], simd=True)
            w_avx = max(w_avx, float((np.abs(g - avx) / np.abs(avx))[strong].max()))
            w_between = max(w_between, float((np.abs(avx - o32) / np.abs(o32))[strong].max()))
            try:
                import ctypes as C
                ramp = np.arange(1023, dtype=np.float32)  # code[k] = k: the resampled "code" IS the chip index
                sh = np.asarray(job["shifts_chips"], np.float32)
                res = [np.empty(n, np.float32) for _ in range(2 * 3)]
                for flavour, fn in enumerate((R.ref_generic_resampler, R.ref_simd_resampler)):
                    rows = (C.POINTER(C.c_float) * 3)(*[res[3 * flavour + t].ctypes.data_as(C.POINTER(C.c_float)) for t in range(3)])
                    fn.restype = None
                    fn.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_uint, C.c_int, C.c_uint]
                    fn(rows, ramp.ctypes.data, C.c_float(job["rem_code_phase_chips"]), C.c_float(job["code_phase_step_chips"]), sh.ctypes.data, 1023, 3, n)
                flips += int(sum(np.count_nonzero(res[t] != res[3 + t]) for t in range(3)))
                flip_samples += 3 * n
            except Exception as e:  # an oracle/_ref built without the protokernel entry points: the figures above still stand
                flips = -1
    print(f"config2 parity on signal taps: worst |gpu-generic|/|generic| = {w_gen:.3e}, |gpu-u_avx|/|u_avx| = {w_avx:.3e}, "
          f"|u_avx-generic|/|generic| = {w_between:.3e}; chips the AVX resampler selects differently from the generic one: {flips} of {flip_samples} tap-samples "
          f"(TOL_REF = {TOL_REF:.1e})")
    # north_star's letter, against the protokernel volk dispatches on this host: within 1e-5 relative on the taps that hold a signal
    if R is not None and R.ref_simd_supported():
        assert w_avx <= TOL_DISPATCH, f"|gpu - u_avx| / |u_avx| = {w_avx:.3e} on signal taps exceeds {TOL_DISPATCH:.0e}"
    assert w_gen <= TOL_REF
    # batched launch == one-by-one launches (no cross-job leakage), and launch-level determinism
    again = b.correlate(jobs)
    assert np.array_equal(out.view(np.float32), again.view(np.float32))
    single = b.correlate(jobs[5:6])
    assert np.allclose(single[0], out[5], rtol=2e-6, atol=1e-3)  # splits>1 path sums in a different order
    b.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_high_dynamics_modes(gpu, mode):
    """high_dyn=1: 7-arg call with the flag (hd resampler + chirped rotator); high_dyn=2: 6-arg overload.
    Against the float64 truth at the tracking bar; against the reference's float32 _generic at 1e-3, the
    reference's own QA tolerance for these kernels (kernel_tests.h:41,88-89: its hd rotator never
    renormalises phase_doppler, K/..high_dynamic_rotator..:73,97)."""
    rng = np.random.default_rng(23 + mode)
    n = 25000
    fs = 25e6
    x = synth_gps_l1_stream(3 * n, fs, [3], [1234.5], [100.25], seed_noise=99)
    codes = [oracle.ca_code(3)]
    b = _bank(gpu, codes)
    b.set_stream_host(x)
    jobs = []
    for i in range(6):
        p = tracking_params_for(fs, rng.uniform(-5000, 5000), rng)
        jobs.append(dict(sample_offset=int(rng.integers(0, 2 * n)), n_samples=n - 7 * i, code_slot=0,
                         shifts_chips=[-0.5, 0.0, 0.5] if i % 2 == 0 else [-0.5, -0.15, 0.0, 0.15, 0.5],
                         phase_rate_step_rad=float(rng.uniform(-2e-9, 2e-9)), code_phase_rate_step_chips=float(rng.uniform(-1e-12, 1e-12)),
                         high_dyn=mode, **p))
    for group in (jobs[0::2], jobs[1::2]):
        out = b.correlate(group)
        for j, job in enumerate(group):
            o32, t64, sabs = oracle_job(codes[0], x, job)
            g = out[j, :len(job["shifts_chips"])]
            assert np.all(scale_err(g, t64, sabs) <= TOL_SCALE_GPU), (job, g, t64)
            assert np.all(np.abs(g - o32) <= 1e-3 * np.abs(o32) + 1e-3 * np.abs(t64).max()), (job, g, o32)
    b.close()


def test_edge_cases_and_errors(gpu):
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.tracking import CorrelatorBank
    x = np.ones(1000, np.complex64)
    b = CorrelatorBank(2, 1023, device=gpu)
    b.set_code(0, oracle.ca_code(1))
    # empty batch is a no-op
    assert b.correlate([]).shape == (0, 8)
    # no stream attached
    with pytest.raises(GshError):
        b.correlate([dict(n_samples=10, shifts_chips=[0.0], code_phase_step_chips=0.1)])
    b.set_stream_host(x)
    # window past the end of the stream
    with pytest.raises(GshError):
        b.correlate([dict(sample_offset=995, n_samples=10, shifts_chips=[0.0], code_phase_step_chips=0.1)])
    # empty code slot, too many taps, mixed modes, descending hd taps
    with pytest.raises(GshError):
        b.correlate([dict(n_samples=10, code_slot=1, shifts_chips=[0.0], code_phase_step_chips=0.1)])
    with pytest.raises(GshError):
        b.correlate([dict(n_samples=10, n_taps=9, shifts_chips=[0.0] * 8, code_phase_step_chips=0.1)])
    with pytest.raises(GshError):
        b.correlate([dict(n_samples=10, shifts_chips=[0.0], code_phase_step_chips=0.1),
                     dict(n_samples=10, shifts_chips=[0.0], code_phase_step_chips=0.1, high_dyn=1)])
    with pytest.raises(GshError):
        b.correlate([dict(n_samples=10, shifts_chips=[0.5, 0.0, -0.5], code_phase_step_chips=0.1, high_dyn=1)])
    # window that ends exactly at the stream end on an odd start: the 16-byte path must not read past it
    out = b.correlate([dict(sample_offset=1, n_samples=999, shifts_chips=[0.0], code_phase_step_chips=0.0, rem_code_phase_chips=0.0)])
    assert out[0, 0] == np.complex64(999 * oracle.ca_code(1)[0])
    b.close()
