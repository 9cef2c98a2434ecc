"""Why the gate against the reference's _generic protokernel is 2.5e-5 and not north_star's 1e-5: what |generic - exact| is made of.

volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn_generic (K/...rotator_dot_prod_32fc_xn.h:66-98) (a) adds its 25 000 products one after the other into ONE float32
accumulator per tap and (b) advances its phasor by one float32 complex product per sample, restoring only its modulus every 256 samples.  Three CPU evaluations
of the same BASELINE config 2 jobs take the two apart:
   A  exact phasor per sample (float64, rounded to float32), float32 products, float32 SEQUENTIAL sums -- the generic kernel's order of summation, no NCO drift;
   B  the reference's _generic kernel itself (oracle/_ref, restated bit for bit in oracle/gnss_oracle.c);
   T  the float64 truth.
Measured here: |A - T| = 5.7e-6 |T| on signal taps, |B - A| (the NCO's share) = 9e-6, |B - T| = 6.9e-6 (the two partly cancel here; 1.14e-5 on the GPU box's config 2 draw).  The order of
summation ALONE puts the generic kernel several 1e-6 from exact arithmetic.  The engine sums in parallel (four accumulator sets per lane, a wave scan, a step over
the waves) with exact re-seeds of its phasor and sits within 1e-6 of T (measured 8e-9 .. 4e-8): it CANNOT also sit within 1e-5 of B on every window unless it
reproduced a sequential float32 sum -- i.e. ran on one lane.  Hence tests/helpers.py: 1e-6 against the truth, TOL_DISPATCH = 1e-5 (north_star's figure) against
the protokernel volk dispatches on an x86 host (_u_avx: eight partial sums per tap, a quarter of the phasor steps; measured 4.9e-6), and TOL_REF = 2.5e-5 against
_generic BY MEASUREMENT of _generic's own distance from exact arithmetic."""
import numpy as np

import oracle
from helpers import TOL_REF, oracle_job, synth_gps_l1_stream, tracking_params_for


def _exact_phasor_float32_sequential(code, x, job):
    """the generic kernel's arithmetic with ONE change: the phasor of sample n is exp(-j (rem + n step)) evaluated in float64 and rounded to float32, not recurred"""
    n = job["n_samples"]
    win = x[job["sample_offset"]:job["sample_offset"] + n]
    sh = np.asarray(job["shifts_chips"], np.float32)
    idx = oracle.code_indices(n, sh, job["rem_code_phase_chips"], job["code_phase_step_chips"], 0.0, len(code), False)
    ph = np.float64(np.float32(job["rem_carr_phase_rad"])) + np.arange(n, dtype=np.float64) * np.float64(np.float32(job["phase_step_rad"]))
    pr, pi = np.cos(ph).astype(np.float32), (-np.sin(ph)).astype(np.float32)
    xr, xi = win.real.astype(np.float32), win.imag.astype(np.float32)
    yr = xr * pr - xi * pi            # float32 products and one float32 subtraction / addition, as std::complex<float> multiplies
    yi = xr * pi + xi * pr
    out = np.zeros(len(sh), np.complex64)
    for t in range(len(sh)):
        c = code[idx[t]].astype(np.float32)
        out[t] = np.cumsum(yr * c, dtype=np.float32)[-1] + 1j * np.cumsum(yi * c, dtype=np.float32)[-1]   # cumsum adds in sequence, in float32
    return out


def test_what_the_generic_kernels_distance_from_exact_arithmetic_is_made_of():
    fs, n = 25e6, 25000
    prns, dops, cphs = [1, 2, 3, 4], [1000.0, -4300.0, 2750.0, -120.0], [10.0, 500.5, 900.25, 333.0]
    x = synth_gps_l1_stream(3 * n, fs, prns, dops, cphs, cn0_dbhz=45.0, seed_noise=3)
    rng = np.random.default_rng(8)
    worst_order = worst_nco = worst_generic = 0.0
    for k, (prn, fd, cph) in enumerate(zip(prns, dops, cphs)):
        code = oracle.ca_code(prn)
        f_code = 1.023e6 * (1 + fd / 1575.42e6)
        start = (1023.0 - cph) / f_code * fs
        off = int(np.ceil(start))
        p = tracking_params_for(fs, fd, rng)
        p["rem_code_phase_chips"] = float(np.float32(-(off - start) * f_code / fs))
        job = dict(sample_offset=off, n_samples=n, shifts_chips=[-0.5, 0.0, 0.5], **p)
        o32, t64, sabs = oracle_job(code, x, job)
        seq = _exact_phasor_float32_sequential(code, x, job)
        strong = np.abs(t64) > 0.01 * sabs
        assert strong[1]                                                     # the prompt holds the signal
        e_order = float((np.abs(seq - t64) / np.abs(t64))[strong].max())
        e_nco = float((np.abs(o32 - seq) / np.abs(t64))[strong].max())
        e_generic = float((np.abs(o32 - t64) / np.abs(t64))[strong].max())
        assert e_generic <= e_order + e_nco + 1e-9
        worst_order, worst_nco, worst_generic = max(worst_order, e_order), max(worst_nco, e_nco), max(worst_generic, e_generic)
    print(f"signal taps, 25 000-sample windows, relative to |truth|: sequential float32 sums with an exact phasor {worst_order:.2e}; the recurred NCO on top {worst_nco:.2e}; "
          f"the reference's _generic kernel {worst_generic:.2e}")
    assert worst_order >= 2e-6          # the order of summation alone is already more than the engine's whole distance from the truth (<= 1e-6) ...
    assert worst_nco >= 5e-7            # ... the NCO adds its drift ...
    assert worst_generic <= TOL_REF     # ... and the gate against _generic leaves room for both
