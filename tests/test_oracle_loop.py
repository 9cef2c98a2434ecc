"""CPU tests of the loop-closure oracle (oracle/gnss_oracle_loop.c) -- discriminators, Tracking_loop_filter,
Tracking_FLL_PLL_filter and the closed DLL/PLL loop -- against
  * the known answers of the reference's own unit tests
    (tests/unit-tests/signal-processing-blocks/tracking/tracking_loop_filter_test.cc:22-206, discriminator_test.cc:35-85),
  * the reference's own objects compiled into oracle/_ref (exact equality; skipped where _ref is absent),
  * golden vectors minted from those objects (tests/golden/loop.npz, tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- tracking_loop_filter_test.cc: impulse responses, bandwidth 5 Hz, T = 1 ms ----------------------------------
@pytest.mark.parametrize("order,last,expected,tol", [
    (1, False, [0.0, 0.0, 20.0, 0.0, 0.0, 0.0], 1e-5),                                # :44-52 (result == input * g1)
    (1, True, [0.0, 0.0, 0.01, 0.02, 0.02, 0.02], 1e-4),                              # :72-82
    (2, False, [0.0, 0.0, 13.37778, 0.0889, 0.0889, 0.0889], 1e-4),                   # :102-112
    (2, True, [0.0, 0.0, 0.006689, 0.013422, 0.013511, 0.013600], 1e-4),              # :132-142
    (3, False, [0.0, 0.0, 15.31877, 0.04494, 0.04520, 0.04546], 1e-4),                # :162-172
    (3, True, [0.0, 0.0, 0.007659, 0.015341, 0.015386, 0.015432], 1e-4),              # :192-202
])
def test_loop_filter_reference_known_answers(order, last, expected, tol):
    out = oracle.loop_filter_run(0.001, 5.0, order, last, 0.0, [0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    assert np.allclose(out, expected, atol=tol), (order, last, out)


def _bpsk_corr(tau):  # discriminator_test.cc:23-32
    return max(0.0, 1.0 - abs(tau))


def test_dll_discriminator_reference_known_answers():
    """DllNcEMinusLNormalizedTest.Bpsk (discriminator_test.cc:35-69): the discriminator returns the code error"""
    L = oracle.lib()
    for a in (1 + 0j, -1 + 0j, 1j, 1 + 1j):
        for spacing in (0.5, 0.25, 0.1, 0.01):
            for err in (0.0, 0.01, 0.1, 0.25, -0.25, -0.1, -0.01):
                e = np.complex64(a * np.float32(_bpsk_corr(err - spacing)))
                l = np.complex64(a * np.float32(_bpsk_corr(err + spacing)))
                d = L.oracle_dll_nc_e_minus_l_normalized(e.real, e.imag, l.real, l.imag, spacing, 1.0, 1.0)
                if abs(err) < 2.0 * spacing:
                    assert abs(d - err) < 1e-4, (a, spacing, err, d)
                else:
                    assert err * d >= 0.0


@pytest.mark.skipif(oracle.ref() is None or not hasattr(oracle.ref(), "ref_fll_diff_atan"), reason="oracle/_ref without loop objects")
def test_loop_oracle_equals_live_reference():
    L, R = oracle.lib(), oracle.ref()
    rng = np.random.default_rng(11)
    v = rng.standard_normal((200, 8)).astype(np.float32) * np.float32(1000.0)
    v[0, :] = 0.0          # zero prompt / zero early+late
    v[1, 0] = 0.0          # I = 0: Costas discriminator returns 0, FLL sees a NaN
    for row in v:
        a = [float(t) for t in row]
        assert L.oracle_pll_cloop_two_quadrant_atan(a[0], a[1]) == R.ref_pll_cloop_two_quadrant_atan(a[0], a[1])
        assert L.oracle_pll_four_quadrant_atan(a[0], a[1]) == R.ref_pll_four_quadrant_atan(a[0], a[1])
        assert L.oracle_fll_diff_atan(a[0], a[1], a[2], a[3], 0.0, 0.001) == R.ref_fll_diff_atan(a[0], a[1], a[2], a[3], 0.0, 0.001)
        assert L.oracle_dll_nc_e_minus_l_normalized(a[0], a[1], a[2], a[3], 0.5, 1.0, 1.0) == R.ref_dll_nc_e_minus_l_normalized(a[0], a[1], a[2], a[3], 0.5, 1.0, 1.0)
        assert L.oracle_dll_nc_e_minus_l_normalized(a[0], a[1], a[2], a[3], 0.15, -2.9, 1.2) == R.ref_dll_nc_e_minus_l_normalized(a[0], a[1], a[2], a[3], 0.15, -2.9, 1.2)
        assert L.oracle_dll_nc_vemlp_normalized(*a) == R.ref_dll_nc_vemlp_normalized(*a)
    x = rng.standard_normal(300).astype(np.float32)
    for order in (1, 2, 3):
        for last in (0, 1):
            for bw, T, init in ((2.0, 0.001, 0.0), (5.0, 0.004, 0.3), (0.75, 0.02, -1.5)):
                out = np.zeros(len(x), np.float32)
                R.ref_loop_filter_run(T, bw, order, last, init, x, out, len(x))
                assert np.array_equal(out, oracle.loop_filter_run(T, bw, order, last, init, x)), (order, last, bw)
    fd = (rng.standard_normal(300) * 3).astype(np.float32)
    pd = (rng.standard_normal(300) * 0.05).astype(np.float32)
    for order in (2, 3):
        for fll, pll, dop, T in ((35.0, 35.0, 1234.5, 0.001), (10.0, 5.0, -3000.0, 0.004), (0.0, 15.0, 0.0, 0.02)):
            out = np.zeros(len(fd), np.float32)
            R.ref_fll_pll_filter_run(fll, pll, order, dop, fd, pd, T, out, len(fd))
            assert np.array_equal(out, oracle.fll_pll_filter_run(fll, pll, order, dop, fd, pd, T)), (order, fll, pll)


def test_loop_oracle_matches_golden():
    z = np.load(os.path.join(HERE, "golden", "loop.npz"))
    L = oracle.lib()
    v = z["disc_inputs"]
    got = np.array([[L.oracle_pll_cloop_two_quadrant_atan(*map(float, r[:2])), L.oracle_fll_diff_atan(*map(float, r[:4]), 0.0, 0.001),
                     L.oracle_dll_nc_e_minus_l_normalized(*map(float, r[:4]), 0.5, 1.0, 1.0), L.oracle_dll_nc_vemlp_normalized(*map(float, r))]
                    for r in v])
    assert np.array_equal(got, z["disc_outputs"])
    for key in [k for k in z.files if k.startswith("lf_out_")]:
        _, _, order, last = key.split("_")
        assert np.array_equal(oracle.loop_filter_run(0.001, 2.0, int(order), int(last), 0.0, z["lf_in"]), z[key]), key
    for key in [k for k in z.files if k.startswith("fp_out_")]:
        order = int(key.split("_")[2])
        assert np.array_equal(oracle.fll_pll_filter_run(35.0, 35.0, order, 1500.0, z["fp_fll"], z["fp_pll"], 0.001), z[key]), key


def test_closed_loop_locks_onto_a_synthetic_signal():
    """A GPS L1 C/A signal at 47 dB-Hz, 1.2 kHz Doppler, handed over with acquisition errors of 20 Hz and 0.2 chip: the
    loop must pull in (mean Doppler within 1 Hz, prompt energy near A*N, mean code error below 0.06 chip) and stay
    there, for the 3rd- and 2nd-order carrier filters and with the FLL pull-in."""
    fs, n, epochs = 4e6, 4000, 500
    fd, cph = 1200.0, 417.3
    x = synth_gps_l1_stream(epochs * n + 2 * n, fs, [9], [fd], [cph], cn0_dbhz=47.0, seed_noise=21)
    f_code = 1.023e6 * (1 + fd / 1575.42e6)
    start_exact = (1023.0 - cph) / f_code * fs
    start = int(round(start_exact + 0.2 * fs / 1.023e6))  # 0.2 chip late
    amp = np.sqrt(10 ** 4.7 * 2.0 / fs) * n
    for kw in (dict(), dict(pll_filter_order=2), dict(enable_fll_pull_in=1, pull_in_time_s=0)):
        conf = oracle.trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=4.0, **kw)
        rec = oracle.trk_run(conf, oracle.ca_code(9), x, start, 0, fd - 20.0, epochs)
        assert len(rec) == epochs
        tail = rec[-100:]
        assert abs(np.mean([r.carrier_doppler_hz for r in tail]) - fd) < 1.0, kw
        p = np.array([complex(r.corr[2], r.corr[3]) for r in tail])
        assert np.mean(np.abs(p)) > 0.9 * amp, kw
        assert abs(np.mean([r.code_error_chips for r in tail])) < 0.06, kw
        # window positions advance by one code period (+-1 sample) and follow the code Doppler
        steps = np.diff([r.sample_counter for r in rec])
        assert set(np.unique(steps)) <= {3999, 4000, 4001}
        assert rec[0].flags == 1  # pull-in transitory: less than pull_in_time_s whole seconds since acquisition


# ---- lock detectors and C/N0: the C restatement against the reference's own objects (oracle/_ref) ---------
def _ref_or_skip():
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_smoother_run"):
        pytest.skip("oracle/_ref (reference build) not present")
    return R


def test_lock_detectors_equal_reference_objects():
    """cn0_m2m4_estimator and carrier_lock_detector (T/lock_detectors.cc:61-133): bit equality with the reference's functions on
    noisy prompts of several C/N0, on noise alone (the NaN branch of :92-100), on zeros, and for the length-1 call of trk.cc:1184."""
    R = _ref_or_skip()
    rng = np.random.default_rng(41)
    for amp in (0.0, 0.3, 1.0, 5.0, 40.0, 1000.0):
        for n in (1, 2, 20, 64):
            p = (amp * rng.choice([-1.0, 1.0], n) + rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            pf = np.ascontiguousarray(p).view(np.float32)
            for t in (0.001, 0.004, 0.02):
                a = oracle.cn0_m2m4_estimator(p, t)
                b = float(R.ref_cn0_m2m4_estimator(pf, n, t))
                assert (np.isnan(a) and np.isnan(b)) or a == b, (amp, n, t, a, b)
            assert oracle.carrier_lock_detector(p) == float(R.ref_carrier_lock_detector(pf, n))
            assert oracle.carrier_lock_detector(p, 1) == float(R.ref_carrier_lock_detector(pf, 1))
    z = np.zeros(20, np.complex64)
    assert oracle.cn0_m2m4_estimator(z, 0.001) == float(R.ref_cn0_m2m4_estimator(z.view(np.float32), 20, 0.001)) == -100.0
    assert oracle.carrier_lock_detector(z) == 0.0
    # a sanity anchor in physical units: 45 dB-Hz over 1 ms -> SNR per sample 10^(4.5) * 1e-3 = 31.6
    amp = np.sqrt(10 ** 4.5 * 1e-3 * 2.0)
    est = [oracle.cn0_m2m4_estimator((amp * rng.choice([-1.0, 1.0], 20) + rng.standard_normal(20) + 1j * rng.standard_normal(20)).astype(np.complex64), 0.001)
           for _ in range(200)]
    assert abs(np.mean(est) - 45.0) < 1.0


def test_exponential_smoother_equals_reference_object():
    """Exponential_Smoother (T/exponential_smoother.cc) as trk.cc:680-692 configures its two instances: initialisation by averaging,
    the flush-and-restart when the average is below min_value + offset (:95-100), then the recursion -- bit equality."""
    R = _ref_or_skip()
    rng = np.random.default_rng(43)
    for alpha, n_init, mn, off in ((0.002, 200, 25.0, 12.0), (0.002, 50, 25.0, 12.0), (0.002, 25, -1.0, 0.0), (0.5, 1, -1.0, 0.0), (1.5, 0, 25.0, 12.0)):
        for base in (10.0, 36.9, 37.1, 45.0, 0.9):
            raw = (base + rng.standard_normal(700)).astype(np.float32)
            raw[300:340] -= 30.0  # a fade
            exp = np.zeros_like(raw)
            R.ref_smoother_run(alpha, n_init, mn, off, raw, len(raw), exp)
            got = oracle.smoother_run(alpha, n_init, raw, mn, off)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (alpha, n_init, mn, off, base)


def test_oracle_loop_lock_detectors_declare_loss_on_noise_only():
    """cn0_and_tracking_lock_status in the loop (trk.cc:1167-1224, 2008-2014): a channel with a signal reports its C/N0 and stays
    locked; a channel correlating noise runs its code-lock fail counter past max_code_lock_fail once the pull-in transitory is
    over and stops with flags bit 1 -- at the period the counter arithmetic predicts."""
    fs, n = 2.046e6, 2046
    epochs = 1200
    x = synth_gps_l1_stream((epochs + 3) * n, fs, [4], [1500.0], [0.0], cn0_dbhz=47.0, seed_noise=5)
    conf = oracle.trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, max_code_lock_fail=50)
    rec = oracle.trk_run(conf, oracle.ca_code(4), x, 0, 0, 1490.0, epochs)
    assert len(rec) == epochs and not any(r.flags & 2 for r in rec)
    tail = [r.cn0_db_hz for r in rec[-200:]]
    assert abs(np.mean(tail) - 47.0) < 2.0, np.mean(tail)
    lt = [r.carrier_lock_test for r in rec[-200:]]       # alpha = 0.002: still converging towards 1 after 1.2 s
    assert np.mean(lt) > conf.carrier_lock_th and lt[-1] > lt[0]
    assert all(r.cn0_db_hz == 0.0 for r in rec[:20]) and rec[20].cn0_db_hz != 0.0   # the first cn0_samples periods only fill the buffer
    # noise only (PRN 9 is not in the stream).  The M2M4 estimate of pure noise hovers around 25-27 dB-Hz (its NaN branch,
    # T/lock_detectors.cc:92-100, gives (E|I|)^2 / (m2 - (E|I|)^2) ~ 0.47 -> 26.7 dB-Hz at 1 ms), so the flag default cn0_min = 25 does
    # not separate it; 32 dB-Hz does
    conf.cn0_min = 32
    rec = oracle.trk_run(conf, oracle.ca_code(9), x, 0, 0, -800.0, epochs)
    assert rec[-1].flags & 2 and len(rec) < epochs
    # pull-in ends at the first period whose start is at least (pull_in_time_s + 1) whole seconds after the acquisition stamp
    first_free = next(i for i, r in enumerate(rec) if not (r.flags & 1))
    assert rec[first_free].sample_counter >= int(fs) > rec[first_free - 1].sample_counter
    # replay of the code-lock counter (trk.cc:1199-1209) from the recorded C/N0 values: up below cn0_min, down (not below 0) otherwise
    cnt, lost_at = 0, None
    for i in range(first_free, len(rec)):
        cnt = cnt + 1 if rec[i].cn0_db_hz < conf.cn0_min else max(cnt - 1, 0)
        if cnt > conf.max_code_lock_fail:
            lost_at = i
            break
    assert lost_at == len(rec) - 1
    assert rec[-1].prn_length_samples == 0


def test_trackstate_restatements_equal_the_golden_vectors():
    """tests/golden/trackstate.npz holds outputs of the reference's own lock_detectors.cc / exponential_smoother.cc / bit_synchronizer.cc
    (minted by tests/golden/make_golden_trackstate.py from oracle/_ref); the C restatements must reproduce them bit for bit -- this pin
    does not need the reference tree at test time."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trackstate.npz"))
    for i, p in enumerate(g["lock_prompts"]):
        for j, t in enumerate((0.001, 0.004, 0.02)):
            a, b = np.float32(oracle.cn0_m2m4_estimator(p, t)), g["lock_cn0_out"][i, j]
            assert (np.isnan(a) and np.isnan(b)) or a == b
        assert np.float32(oracle.carrier_lock_detector(p)) == g["lock_detector_out"][i, 0]
        assert np.float32(oracle.carrier_lock_detector(p, 1)) == g["lock_detector_out"][i, 1]
    k = 0
    for (alpha, n_init, mn, off) in g["smoother_cfg"]:
        for _ in range(5):
            got = oracle.smoother_run(float(alpha), int(n_init), g["smoother_raw"][k], float(mn), float(off))
            assert np.array_equal(got.view(np.uint32), g["smoother_out"][k].view(np.uint32)), k
            k += 1
    for cfg, p, ok, ev, un in zip(g["bitsync_cfg"], g["bitsync_prompts"], g["bitsync_ok"], g["bitsync_event_out"], g["bitsync_until_edge_out"]):
        e2, u2 = oracle.bit_sync_run(p, int(cfg[0]), int(cfg[1]), int(cfg[2]), float(cfg[3]), float(cfg[4]), bool(cfg[5]), quality_ok=ok)
        assert np.array_equal(e2, ev) and np.array_equal(u2, un)
