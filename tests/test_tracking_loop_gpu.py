"""GPU tests of the device-resident DLL/PLL loop (gsh_trk_*) against the CPU oracle loop (oracle.trk_run).

The loop has feedback, so the comparison is a trajectory comparison: correlator outputs differ from the float32 oracle at
the 1e-7..1e-5 level (tests/test_tracking_gpu.py), libm differs from the device math library in the last ulp, and both
are fed back through the loop filters.  Because the fed-back float32 code phase can differ in its last bit, a sample that
sits exactly on a chip edge may pick the neighbouring chip on one side (the accumulator then moves by 2|x[n]|); see
_compare for how the bars change after the first such event.  Bars before it: every period's window position identical (a +-1-sample difference may only
appear where the float64 block length sits within 1e-6 of an integer -- asserted), Doppler within 0.05 Hz, code
frequency within 2e-3 chip/s, correlator outputs within 2e-4 of the prompt magnitude, discriminator outputs within 1e-4.
"""
import numpy as np
import pytest

import oracle
from helpers import add_code_signal, cn0_to_amplitude, golden_e1_l5_codes, synth_gps_l1_stream

pytestmark = pytest.mark.gpu


def _loop(gpu, conf_kw, n_channels, max_len):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    return TrackingLoop(trk_conf(**conf_kw), n_channels, max_len, device=gpu)


def _compare(rec_gpu, rec_ora, n_taps, tag, xmax=6.0):
    """Trajectory comparison.  Until the first chip-edge flip the two loops must agree tightly; a flip is a one-sample
    perturbation (2|x[n]| in one tap) that the two loops then filter slightly differently: their code phases differ by
    ~1e-4 chip afterwards, so about one edge sample per period disagrees from then on and the bars become those of two
    loops tracking the same signal from nearly identical states (a few samples' worth on the correlators)."""
    assert len(rec_gpu) == len(rec_ora), (tag, len(rec_gpu), len(rec_ora))
    flips = 0
    shifted_from = None  # first period whose window start differs by one sample (block length rounded across an integer boundary)
    compared = 0
    pi = n_taps - 1 if n_taps == 3 else 4  # index of the prompt's real part in corr[]
    for e, (g, o) in enumerate(zip(rec_gpu, rec_ora)):
        near_int = min(o.rem_code_phase_samples, 1.0 - o.rem_code_phase_samples)
        if shifted_from is None and g.sample_counter != o.sample_counter:
            # only legitimate when an earlier block length sat on an integer boundary (or after a flip nudged it there)
            prev = rec_ora[e - 1]
            assert min(prev.rem_code_phase_samples, 1.0 - prev.rem_code_phase_samples) < (1e-6 if flips == 0 else 1e-3), (tag, e, g.sample_counter, o.sample_counter)
            shifted_from = e
        if shifted_from is not None:
            # windows offset by one sample from here on: the two loops see different samples, so only what both must agree on as
            # trackers of the same signal is compared -- but it IS compared, to the end of the run
            assert abs(int(g.sample_counter) - int(o.sample_counter)) <= 1, (tag, e, g.sample_counter, o.sample_counter)
            assert abs(g.carrier_doppler_hz - o.carrier_doppler_hz) <= 0.5, (tag, e, g.carrier_doppler_hz, o.carrier_doppler_hz)
            assert abs(g.code_freq_chips - o.code_freq_chips) <= 0.3, (tag, e, g.code_freq_chips, o.code_freq_chips)
            po = np.array(list(o.corr)[:2 * n_taps])
            pg = np.array(list(g.corr)[:2 * n_taps])
            scale = max(np.hypot(po[pi], po[pi + 1]), 50.0)
            assert abs(np.hypot(pg[pi], pg[pi + 1]) - np.hypot(po[pi], po[pi + 1])) <= 8.0 * xmax + 1e-2 * scale, (tag, e, pg, po)
            compared += 1
            continue
        assert g.flags == o.flags, (tag, e)
        pg = np.array(list(g.corr)[:2 * n_taps])
        po = np.array(list(o.corr)[:2 * n_taps])
        scale = max(np.hypot(po[pi], po[pi + 1]), 50.0)
        dev = np.max(np.abs(pg - po))
        if dev > 2e-4 * scale:
            flips += 1  # from here on the two code phases differ in their last bits: edge samples disagree routinely
        loose = flips > 0
        if loose:
            assert dev <= 4.0 * xmax + 3e-3 * scale, (tag, e, pg, po)
        assert abs(g.carrier_doppler_hz - o.carrier_doppler_hz) <= (0.5 if loose else 0.05), (tag, e, g.carrier_doppler_hz, o.carrier_doppler_hz)
        assert abs(g.code_freq_chips - o.code_freq_chips) <= (0.3 if loose else 2e-3), (tag, e, g.code_freq_chips, o.code_freq_chips)
        assert abs(g.code_error_chips - o.code_error_chips) <= (8.0 * xmax / scale if loose else 1e-4), (tag, e)
        assert abs(g.carr_phase_error_hz - o.carr_phase_error_hz) <= (8.0 * xmax / scale if loose else 1e-4), (tag, e)
        assert g.prn_length_samples == o.prn_length_samples or near_int < (1e-3 if loose else 1e-6), (tag, e)
        if not loose:
            assert abs(g.rem_code_phase_samples - o.rem_code_phase_samples) <= 1e-4 or near_int < 1e-4, (tag, e)
        assert abs(g.acc_carrier_phase_rad - o.acc_carrier_phase_rad) <= (5e-2 if loose else 1e-3) * max(1.0, abs(o.acc_carrier_phase_rad)), (tag, e)
        compared += 1
    # every period is held to one of the two sets of bars; nothing is silently skipped
    assert compared == len(rec_ora) and compared >= 0.9 * len(rec_ora), (tag, compared, len(rec_ora), shifted_from)
    return dict(compared=compared, shifted_from=shifted_from, flips=flips)


@pytest.mark.parametrize("variant", ["pll3", "pll2_fll", "no_aiding_dll1"])
def test_gps_l1_closed_loop_matches_oracle_and_locks(gpu, variant):
    fs, n, epochs = 4e6, 4000, 300
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=4.0)
    if variant == "pll2_fll":
        kw.update(pll_filter_order=2, enable_fll_pull_in=1, enable_fll_steady_state=1, fll_bw_hz=10.0, pull_in_time_s=0)
    if variant == "no_aiding_dll1":
        kw.update(carrier_aiding=0, dll_filter_order=1, dll_bw_hz=1.0, cloop=0)
    prns = [3, 9, 17, 22]
    dops = [1200.0, -2750.0, 4100.0, 35.0]
    cphs = [417.3, 12.9, 800.4, 333.3]
    x = synth_gps_l1_stream(epochs * n + 3 * n, fs, prns, dops, cphs, cn0_dbhz=47.0, seed_noise=31)
    loop = _loop(gpu, kw, n_channels=5, max_len=1023)   # channel 4 is never started
    loop.set_stream_host(x)
    conf_o = oracle.trk_conf(**kw)
    starts = []
    for ch, (prn, fd, cph) in enumerate(zip(prns, dops, cphs)):
        f_code = 1.023e6 * (1 + fd / 1575.42e6)
        start = int(round((1023.0 - cph) / f_code * fs + 0.15 * fs / 1.023e6))  # 0.15 chip late
        starts.append(start)
        loop.start(ch, oracle.ca_code(prn), start, 0, fd - 12.0)
    rec, done = loop.run(epochs)
    assert done[4] == 0 and len(rec[4]) == 0
    amp = cn0_to_amplitude(47.0, fs) * n
    for ch, (prn, fd) in enumerate(zip(prns, dops)):
        ora = oracle.trk_run(conf_o, oracle.ca_code(prn), x, starts[ch], 0, fd - 12.0, epochs)
        assert done[ch] == epochs == len(ora)
        _compare(rec[ch], ora, 3, f"{variant} ch{ch}")
        tail = rec[ch][-80:]
        assert abs(np.mean([r.carrier_doppler_hz for r in tail]) - fd) < 1.5, (variant, ch)
        if variant != "no_aiding_dll1":
            # (a first-order 1 Hz code loop without carrier aiding keeps a steady-state error of code-Doppler / 4 chips:
            #  up to 0.67 chip at 4.1 kHz -- agreement with the oracle is the only claim for that variant)
            assert np.mean([np.hypot(r.corr[2], r.corr[3]) for r in tail]) > 0.9 * amp, (variant, ch)
    # a second call continues from the device-resident state: identical to one long run
    rec2, done2 = loop.run(50)
    ora_long = oracle.trk_run(conf_o, oracle.ca_code(prns[0]), x, starts[0], 0, dops[0] - 12.0, epochs + 50)
    n_more = len(ora_long) - epochs
    assert done2[0] == n_more
    _compare(rec2[0], ora_long[epochs:], 3, f"{variant} continuation")
    loop.close()


def test_galileo_e1_veml_pilot_and_data(gpu):
    """E1: 4 ms code period, 32 Msps -> N = 128 000, VE/E/P/L/VL on E1C + data prompt on E1B, sinBOC replica at 2 samples
    per chip, VEMLP discriminator, four-quadrant PLL discriminator (pilot)."""
    fs, n, epochs = 32e6, 128000, 40
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(41)
    n_stream = (epochs + 2) * n
    x = (rng.standard_normal(n_stream) + 1j * rng.standard_normal(n_stream)).astype(np.complex64)
    amp = cn0_to_amplitude(45.0, fs)
    fd, ph = -1830.0, 3000.0
    rate = 1.023e6 * (1 + fd / 1575.42e6) / fs * 2.0
    add_code_signal(x, (g["e1b"][7] - g["e1c"][7]) / np.sqrt(2.0), fs, rate, ph, fd, amp)
    kw = dict(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, cloop=0,
              early_late_space_chips=0.15, very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pll_filter_order=3, dll_filter_order=2)
    loop = _loop(gpu, kw, n_channels=2, max_len=8184)
    loop.set_stream_host(x)
    start = int(round((8184.0 - ph) / rate))
    loop.start(0, g["e1c"][7], start, 0, fd - 5.0, data_code=g["e1b"][7])
    loop.start(1, g["e1c"][20], start + 777, 0, 950.0, data_code=g["e1b"][20])  # no such signal: noise-driven loop
    rec, done = loop.run(epochs)
    conf_o = oracle.trk_conf(**kw)
    for ch, (prn_i, st, dop) in enumerate(((7, start, fd - 5.0), (20, start + 777, 950.0))):
        ora = oracle.trk_run(conf_o, g["e1c"][prn_i], x, st, 0, dop, epochs, data_code=g["e1b"][prn_i])
        assert done[ch] == len(ora)
        _compare(rec[ch], ora, 5, f"e1 ch{ch}")
        for rg, ro in zip(rec[ch], ora):
            assert abs(rg.prompt_data[0] - ro.prompt_data[0]) <= 2e-4 * max(50.0, abs(complex(*ro.prompt_data)))
    # the tracked channel holds the signal: pilot and data prompts both carry A*N/sqrt(2)
    tail = rec[0][-10:]
    expect = amp * n / np.sqrt(2.0)
    assert np.mean([np.hypot(r.corr[4], r.corr[5]) for r in tail]) > 0.85 * expect
    assert np.mean([np.hypot(*r.prompt_data) for r in tail]) > 0.85 * expect
    loop.close()


def test_loop_errors_and_stream_end(gpu):
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    with pytest.raises(GshError):
        TrackingLoop(trk_conf(pll_filter_order=4), 1, 1023, device=gpu)
    loop = TrackingLoop(trk_conf(), 1, 1023, device=gpu)
    with pytest.raises(GshError):
        loop.run(1)  # no stream
    x = synth_gps_l1_stream(10 * 4000 + 100, 4e6, [1], [500.0], [0.0], cn0_dbhz=50.0, seed_noise=3)
    loop.set_stream_host(x)
    rec, done = loop.run(5)
    assert done == [0]  # nothing started
    loop.start(0, oracle.ca_code(1), 10, 0, 500.0)
    rec, done = loop.run(100)  # only 10 windows fit
    assert 9 <= done[0] <= 10 and len(rec[0]) == done[0]
    assert rec[0][-1].sample_counter + 4000 <= len(x)
    loop.close()


def test_lock_detectors_and_cn0_on_device(gpu):
    """cn0_and_tracking_lock_status (trk.cc:1167-1224) inside the device loop.  Channels with a signal report the
    oracle loop's C/N0 and carrier-lock values and stay locked; a channel correlating noise is dropped (flags bit 1, channel
    inactive afterwards) at the same period as in the oracle loop, give or take the periods where its noisy C/N0 estimate sits
    within float rounding of cn0_min."""
    fs, n, epochs = 2.046e6, 2046, 1150
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, cn0_min=32,
              max_code_lock_fail=50)
    prns, dops, cphs = [4, 12], [1500.0, -2200.0], [0.0, 511.5]
    x = synth_gps_l1_stream((epochs + 3) * n, fs, prns, dops, cphs, cn0_dbhz=47.0, seed_noise=5)
    loop = _loop(gpu, kw, n_channels=3, max_len=1023)
    loop.set_stream_host(x)
    conf_o = oracle.trk_conf(**kw)
    starts = []
    for ch, (prn, fd, cph) in enumerate(zip(prns, dops, cphs)):
        f_code = 1.023e6 * (1 + fd / 1575.42e6)
        start = int(round(((1023.0 - cph) % 1023.0) / f_code * fs))
        starts.append(start)
        loop.start(ch, oracle.ca_code(prn), start, 0, fd - 8.0)
    loop.start(2, oracle.ca_code(9), 0, 0, -800.0)          # PRN 9 is not in the stream
    rec, done = loop.run(epochs)
    for ch, (prn, fd) in enumerate(zip(prns, dops)):
        ora = oracle.trk_run(conf_o, oracle.ca_code(prn), x, starts[ch], 0, fd - 8.0, epochs)
        assert done[ch] == len(ora) == epochs and not any(r.flags & 2 for r in rec[ch])
        g_cn0 = np.array([r.cn0_db_hz for r in rec[ch]])
        o_cn0 = np.array([r.cn0_db_hz for r in ora])
        assert np.all(g_cn0[:20] == 0.0) and g_cn0[20] != 0.0
        # the estimator sees the prompts of both loops, which agree to ~1e-4 of their magnitude: C/N0 within a few hundredths of a dB
        assert np.max(np.abs(g_cn0 - o_cn0)) < 0.25, (ch, np.max(np.abs(g_cn0 - o_cn0)))
        assert abs(np.mean(g_cn0[-200:]) - 47.0) < 2.0
        g_lt = np.array([r.carrier_lock_test for r in rec[ch]])
        o_lt = np.array([r.carrier_lock_test for r in ora])
        assert np.max(np.abs(g_lt - o_lt)) < 2e-2
        assert np.mean(g_lt[-200:]) > 0.7
    ora = oracle.trk_run(conf_o, oracle.ca_code(9), x, 0, 0, -800.0, epochs)
    assert ora[-1].flags & 2 and len(ora) < epochs
    assert done[2] < epochs and rec[2][-1].flags & 2 and rec[2][-1].prn_length_samples == 0
    assert abs(done[2] - len(ora)) <= 12, (done[2], len(ora))
    # the counter arithmetic, replayed from the device's own C/N0 record, predicts the device's own loss period exactly
    first_free = next(i for i, r in enumerate(rec[2]) if not (r.flags & 1))
    cnt, lost_at = 0, None
    for i in range(first_free, len(rec[2])):
        cnt = cnt + 1 if rec[2][i].cn0_db_hz < kw["cn0_min"] else max(cnt - 1, 0)
        if cnt > kw["max_code_lock_fail"]:
            lost_at = i
            break
    assert lost_at == len(rec[2]) - 1
    # a dropped channel stays dropped; the others continue
    rec2, done2 = loop.run(2)
    assert done2[2] == 0 and done2[0] == 2 and done2[1] == 2
    loop.close()


def test_stamp_ahead_of_the_start_keeps_the_pull_in_transitory_over(gpu):
    """trk.cc:1910-1917: d_pull_in_transitory is a latch.  A channel whose acquisition stamp lies beyond its first sample had the latch released at the pull-in call (the
    unsigned difference wrapped); it must not come back when the window position passes the stamp (gsh_trk_start_flags sets GSH_TRK_START_PULL_IN_OVER by itself,
    as the oracle does -- pinned to the reference block in tests/test_oracle_loop_pinned.py).  FLL pull-in on: a returning transitory would bend the carrier loop."""
    fs, n, epochs = 4e6, 4000, 120
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=30.0, dll_bw_hz=2.0, pull_in_time_s=1, enable_fll_pull_in=1, fll_bw_hz=10.0, enable_lock_detectors=1,
              bit_synchronization_time_limit_s=70)
    prn, fd, cph = 7, -2100.0, 700.25
    x = synth_gps_l1_stream((epochs + 10) * n, fs, [prn], [fd], [cph], cn0_dbhz=46.0, seed_noise=23)
    start = int(round((1023.0 - cph) / (1.023e6 * (1 + fd / 1575.42e6)) * fs))
    stamp = start + 4 * n + 77
    loop = _loop(gpu, kw, n_channels=2, max_len=1023)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(prn), start, stamp, fd + 25.0)                       # no flag passed: the engine sees stamp > start
    loop.start(1, oracle.ca_code(prn), start, stamp, fd + 25.0, pull_in_over=True)    # what Hip_Tracking_Runtime passes
    rec, done = loop.run(epochs)
    ora = oracle.trk_run(oracle.trk_conf(**kw), oracle.ca_code(prn), x, start, stamp, fd + 25.0, epochs)
    assert rec[0][10].sample_counter > stamp
    for ch in (0, 1):
        assert done[ch] == len(ora)
        assert not any(r.flags & 1 for r in rec[ch][:done[ch]])
        _compare(rec[ch], ora, 3, f"stamp ahead ch{ch}")
    assert bytes(memoryview(rec[0][-1])) == bytes(memoryview(rec[1][-1]))
    loop.close()


@pytest.mark.parametrize("cn0_samples", [1, 7, 16, 17, 33, 64])
def test_cn0_estimator_over_buffers_of_every_shape(gpu, cn0_samples):
    """The M2M4 sums are formed by the 64 lanes of one wave in rows of sixteen (csrc/tracking_loop.hip, m2m4_sums_wave): buffer lengths inside one row, exactly one row,
    one element into the next row, three rows and all four -- C/N0 and carrier lock test along the oracle loop's, and the carrier-lock half's own copy of the buffer's
    first element (it is what carrier_lock_detector(buffer, 1) looks at, trk.cc:1184) with buffers of one element and of many."""
    fs, n, epochs = 2.046e6, 2046, 260
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, cn0_samples=cn0_samples, cn0_min=25)
    x = synth_gps_l1_stream((epochs + 3) * n, fs, [4], [1500.0], [0.0], cn0_dbhz=47.0, seed_noise=5)
    loop = _loop(gpu, kw, n_channels=1, max_len=1023)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(4), 0, 0, 1492.0)
    rec, done = loop.run(epochs)
    loop.close()
    ora = oracle.trk_run(oracle.trk_conf(**kw), oracle.ca_code(4), x, 0, 0, 1492.0, epochs)
    assert int(done[0]) == len(ora) == epochs
    g_cn0, o_cn0 = np.array([r.cn0_db_hz for r in rec[0]]), np.array([r.cn0_db_hz for r in ora])
    g_lt, o_lt = np.array([r.carrier_lock_test for r in rec[0]]), np.array([r.carrier_lock_test for r in ora])
    assert np.all(g_cn0[:cn0_samples] == 0.0) and np.array_equal(g_cn0 == 0.0, o_cn0 == 0.0)
    # (ONE prompt: the estimator's denominator m_2 - sqrt(2 m_2^2 - m_4) is exactly zero with a correctly rounded square root, hence -100 dB-Hz in every period -- on the
    #  device too since its estimator takes sqrtf instead of __fsqrt_rn, which hipcc renders as a bare v_sqrt_f32, one ulp off now and then)
    finite = np.isfinite(o_cn0) & np.isfinite(g_cn0)
    assert np.array_equal(np.isfinite(o_cn0), np.isfinite(g_cn0))
    tol = 0.25 if cn0_samples >= 16 else 1.5  # (a buffer of a few prompts gives a wild estimate: the same wild one on both sides)
    assert np.max(np.abs(g_cn0[finite] - o_cn0[finite])) < tol, (cn0_samples, np.max(np.abs(g_cn0[finite] - o_cn0[finite])))
    assert np.max(np.abs(g_lt - o_lt)) < 2e-2


def test_carrier_lock_counter_drops_a_noise_channel(gpu):
    """The two halves of cn0_and_tracking_lock_status run on two waves of the loop kernel and thread 0 waits for their verdict only in a period in which a fail
    counter can pass its limit (csrc/tracking_loop.hip, SerialMail).  Here the CARRIER counter is the one that passes it (the code limit is out of reach): on noise
    the smoothed carrier lock test stays far below carrier_lock_th, so every period after the pull-in transitory fails and the channel is dropped
    max_carrier_lock_fail + 1 periods later -- in the same period as in the oracle loop; then the same with the code counter, with both, and with the symbol
    synchronisation running."""
    fs, n, epochs = 4e6, 4000, 1100
    x = synth_gps_l1_stream((epochs + 3) * n, fs, [], [], [], seed_noise=21)
    for kw in (dict(max_carrier_lock_fail=20, max_code_lock_fail=1 << 30), dict(max_carrier_lock_fail=1 << 30, max_code_lock_fail=20, cn0_min=40),
               dict(max_carrier_lock_fail=20, max_code_lock_fail=20, cn0_min=40, enable_symbol_sync=1, symbols_per_bit=20)):
        conf = dict(dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, cn0_min=35), **kw)
        loop = _loop(gpu, conf, n_channels=1, max_len=1023)
        loop.set_stream_host(x)
        loop.start(0, oracle.ca_code(3), 100, 0, 500.0)
        rec, done = loop.run(epochs)
        loop.close()
        ora = oracle.trk_run(oracle.trk_conf(**conf), oracle.ca_code(3), x, 100, 0, 500.0, epochs)
        assert ora[-1].flags & 2 and len(ora) < epochs, kw
        assert int(done[0]) == len(ora) and rec[0][int(done[0]) - 1].flags & 2, (kw, int(done[0]), len(ora))
        first_free = next(i for i, r in enumerate(rec[0]) if not (r.flags & 1))
        assert int(done[0]) - 1 == first_free + 20, (kw, int(done[0]), first_free)  # the twenty-first fail after the pull-in transitory


def test_loop_follows_a_live_ring(gpu):
    """gsh_trk_set_stream_ring: the loop works through whatever the ring holds at each gsh_trk_run call and continues after the next
    push; the ring is much shorter than the stream (it wraps many times) and is fed with 8-bit items.  Same kernel, same arithmetic,
    only the window addresses differ: the records must equal those of one run over the flat converted stream BIT FOR BIT."""
    import ctypes as C
    from gnss_sdr_amd.sample_stream import SampleStream
    from gnss_sdr_amd._lib import TrkEpoch
    fs, n, epochs = 4e6, 4000, 260
    prns, dops, cphs = [5, 18], [-1900.0, 3300.0], [100.0, 900.5]
    total = (epochs + 3) * n
    x = synth_gps_l1_stream(total, fs, prns, dops, cphs, cn0_dbhz=47.0, seed_noise=77)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 25.0), -127, 127).astype(np.int8)
    xf = (x8[:, 0].astype(np.float32) + 1j * x8[:, 1].astype(np.float32)).astype(np.complex64)
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=3.0, enable_lock_detectors=1)
    starts = []
    for fd, cph in zip(dops, cphs):
        f_code = 1.023e6 * (1 + fd / 1575.42e6)
        starts.append(int(round((1023.0 - cph) / f_code * fs)))
    # reference run: flat buffer
    flat = _loop(gpu, kw, n_channels=2, max_len=1023)
    flat.set_stream_host(xf)
    for ch in range(2):
        flat.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    rec_flat, done_flat = flat.run(epochs + 10)   # to the end of the stream
    flat.close()
    # live run: ring of 23 periods + 7 samples, blocks of 9.5 periods
    ring = SampleStream(23 * n + 7, 2 * n, device=gpu)
    live = _loop(gpu, kw, n_channels=2, max_len=1023)
    live.set_stream_ring(ring)
    for ch in range(2):
        live.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    rec_live = [[], []]
    pushed, blk = 0, 9 * n + n // 2
    n_calls = 0
    while pushed < total:
        m = min(blk, total - pushed)
        ring.push(x8[pushed:pushed + m], "ibyte")
        pushed += m
        rec, done = live.run(12)            # more than a block holds: the loop must stop at the newest sample by itself
        n_calls += 1
        for ch in range(2):
            assert done[ch] <= 10
            rec_live[ch] += rec[ch]
    assert n_calls > 25
    for ch in range(2):
        assert len(rec_live[ch]) == done_flat[ch] == len(rec_flat[ch])
        a = b"".join(bytes(memoryview(r)) for r in rec_live[ch])
        b = b"".join(bytes(memoryview(r)) for r in rec_flat[ch])
        assert a == b, f"channel {ch}: live-ring records differ from the flat-buffer run"
        assert abs(np.mean([r.carrier_doppler_hz for r in rec_live[ch][-60:]]) - dops[ch]) < 2.0
    # a channel that falls further behind than the ring holds stops instead of reading overwritten samples
    late = _loop(gpu, kw, n_channels=1, max_len=1023)
    late.set_stream_ring(ring)
    late.start(0, oracle.ca_code(prns[0]), starts[0], 0, dops[0])
    rec, done = late.run(5)
    assert done[0] == 0
    late.close()
    live.close()
    ring.close()


@pytest.mark.gpu
def test_records_written_by_the_kernel_into_host_memory_equal_the_copied_ones(gpu, tmp_path):
    """gsh_trk_run_begin hands the kernel the page-locked host buffers themselves (the default since the end of round 3; GSH_TRK_HOST_RECORDS=0: device
    buffers and two copies behind the kernel).  Both ways must deliver the same bytes: profiles/ab/r03/loop_records.py (8 channels, lock detectors, FLL
    pull-in, channels that lose lock; 4 800 periods in six launches) in one process per mode."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for mode in ("0", "1"):
        path = str(tmp_path / f"records_{mode}.bin")
        r = subprocess.run([sys.executable, os.path.join(root, "profiles", "ab", "r03", "loop_records.py"), path], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GSH_TRK_HOST_RECORDS=mode), cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out.append(open(path, "rb").read())
    assert len(out[0]) > 1_000_000 and out[0] == out[1]


@pytest.mark.parametrize("groups", [2, 3, 4])
def test_cooperating_work_groups_hold_the_oracles_bars(gpu, groups):
    """gsh_trk_set_split: several work-groups share every window of a channel (launched runs, standard correlator).  The partial sums are
    added in rank order -- another order of summation than the one-work-group form's -- so the claim is the SAME bars against the pinned oracle
    loop (not identity with the one-work-group records), at the headline shape (25 Msps: 7 trips per window, so every work-group has a segment)
    and at a short window (4 Msps: one trip -- the helpers' segments are empty and they only hand zeros over)."""
    from gnss_sdr_amd import GshError
    for fs, n, epochs, bw in ((25e6, 25000, 60, 2.0), (4e6, 4000, 120, 4.0)):
        kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=bw, enable_lock_detectors=1)
        prns, dops, cphs = [5, 12, 30], [2300.0, -4100.0, 150.0], [100.2, 640.7, 1001.9]
        x = synth_gps_l1_stream((epochs + 3) * n, fs, prns, dops, cphs, cn0_dbhz=46.0, seed_noise=77)
        loop = _loop(gpu, kw, n_channels=4, max_len=1023)  # channel 3 never started: its work-groups leave at once, helpers included
        loop.set_stream_host(x)
        loop.set_split(groups)
        conf_o = oracle.trk_conf(**kw)
        starts = []
        for ch, (prn, fd, cph) in enumerate(zip(prns, dops, cphs)):
            f_code = 1.023e6 * (1 + fd / 1575.42e6)
            starts.append(int(round((1023.0 - cph) / f_code * fs)) + ch * 17)
            loop.start(ch, oracle.ca_code(prn), starts[-1], 0, fd + 8.0)
        rec, done = loop.run(epochs // 2)
        rec2, done2 = loop.run(epochs - epochs // 2)   # a second launch: the tags of the hand-over words go on, nothing stale is taken
        assert done[3] == 0 and done2[3] == 0
        for ch, (prn, fd) in enumerate(zip(prns, dops)):
            ora = oracle.trk_run(conf_o, oracle.ca_code(prn), x, starts[ch], 0, fd + 8.0, epochs)
            assert done[ch] + done2[ch] == epochs == len(ora)
            _compare(rec[ch] + rec2[ch], ora, 3, f"split{groups} fs{fs:g} ch{ch}")
        # switching back gives the one-work-group form again (and frees the hand-over words)
        loop.set_split(1)
        rec3, done3 = loop.run(2)
        assert done3[:3] == [2, 2, 2]
        loop.close()
    # what the switch refuses: the high-dynamics correlator, more work-groups than the device can hold at once
    hd = _loop(gpu, dict(fs_in=4e6, vector_length=4000, high_dyn=1), n_channels=1, max_len=1023)
    with pytest.raises(GshError):
        hd.set_split(2)
    hd.close()
    big = _loop(gpu, dict(fs_in=4e6, vector_length=4000), n_channels=200, max_len=1023)
    with pytest.raises(GshError):
        big.set_split(8)
    big.close()


def test_cooperating_work_groups_with_the_pilot_and_data_taps(gpu):
    """the VE/E/P/L/VL + data-prompt form (six sums per hand-over) over two and four work-groups"""
    fs, n, epochs = 32e6, 128000, 24
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(43)
    n_stream = (epochs + 2) * n
    x = (rng.standard_normal(n_stream) + 1j * rng.standard_normal(n_stream)).astype(np.complex64)
    amp = cn0_to_amplitude(45.0, fs)
    fd, ph = 2210.0, 5000.0
    rate = 1.023e6 * (1 + fd / 1575.42e6) / fs * 2.0
    add_code_signal(x, (g["e1b"][11] - g["e1c"][11]) / np.sqrt(2.0), fs, rate, ph, fd, amp)
    kw = dict(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, cloop=0,
              early_late_space_chips=0.15, very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pll_filter_order=3, dll_filter_order=2)
    start = int(round((8184.0 - ph) / rate))
    conf_o = oracle.trk_conf(**kw)
    ora = oracle.trk_run(conf_o, g["e1c"][11], x, start, 0, fd - 5.0, epochs, data_code=g["e1b"][11])
    for groups in (2, 4, 0):   # 0: chosen by the library (four for a 62-trip window on an otherwise idle device)
        loop = _loop(gpu, kw, n_channels=1, max_len=8184)
        loop.set_stream_host(x)
        loop.set_split(groups)
        loop.start(0, g["e1c"][11], start, 0, fd - 5.0, data_code=g["e1b"][11])
        rec, done = loop.run(epochs)
        assert done[0] == len(ora)
        _compare(rec[0], ora, 5, f"e1 split{groups}")
        for rg, ro in zip(rec[0], ora):
            assert abs(rg.prompt_data[0] - ro.prompt_data[0]) <= 2e-4 * max(50.0, abs(complex(*ro.prompt_data)))
        loop.close()
