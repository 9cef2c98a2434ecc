"""CPU tests of the Tong and Galileo 8 ms detector oracles (oracle/pcps_oracle.py: TongOracle, Galileo8msOracle) on the
reference's own synthetic test cases (gps_l1_ca_pcps_tong_acquisition_gsoc2013_test.cc,
galileo_e1_pcps_8ms_ambiguous_acquisition_gsoc2013_test.cc): delay error < 0.5 chip, Doppler error < 2 / (3 T)."""
import numpy as np

from oracle.pcps_oracle import CccwsrOracle, FineDopplerOracle, Galileo8msOracle, QuickSyncOracle, TongOracle, count_doppler_bins, mean_input_power
from detector_cases import cccwsr_case, e1_8ms_case, fine_doppler_case, quicksync_case, tong_case


def test_bin_count_is_inclusive():
    assert count_doppler_bins(10000, 250) == 81  # the main PCPS block has 80 (acq.cc:113)
    assert count_doppler_bins(5000, 500) == 21
    assert count_doppler_bins(100, 250) == 1


def test_tong_known_answer_and_counter():
    x, kw, code = tong_case()
    o = TongOracle(**kw)
    o.set_local_code(code)
    counts, states = [], []
    k = 0
    while o.state == 1:
        o.work(x[k * 4000:(k + 1) * 4000])
        counts.append(o.tong_count)
        states.append(o.state)
        k += 1
    # every dwell exceeds threshold * dwell_count: the counter climbs 1 -> 8 in 7 dwells (tong.cc:279-286)
    assert counts == [2, 3, 4, 5, 6, 7, 8] and states[-1] == 2 and o.dwell_count == 7
    assert abs(600.0 - o.result["acq_delay_samples"] * 1023.0 / 4000.0) < 0.5
    assert abs(o.result["doppler_hz"] - 750.0) < 2.0 / 3e-3
    # the statistic is the fraction of the block's power the replica captures, accumulated over the dwells
    assert 7 * 0.00108 < float(o.test_statistics) < 7 * 0.02
    assert float(o.weight) == float(np.float32(1.0) / (np.float32(16e6) * np.float32(16e6) * o.input_power))


def test_tong_noise_only_goes_negative():
    x, kw, code = tong_case(signal=False, seed=5)
    o = TongOracle(**dict(kw, threshold=0.004))
    o.set_local_code(code)
    o.work(x[:4000])
    assert o.state == 3 and o.tong_count == 0 and o.dwell_count == 1  # 1 -> 0 at the first miss (tong.cc:288-294)
    # a generous threshold keeps it counting up on noise until tong_max_dwells stops it (tong.cc:296-299)
    o = TongOracle(**dict(kw, threshold=1e-5, tong_max_val=50, tong_max_dwells=4))
    o.set_local_code(code)
    for k in range(4):
        st = o.work(x[k * 4000:(k + 1) * 4000])
    assert st == 3 and o.dwell_count == 4 and o.tong_count == 5


def test_mean_input_power():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(4000) + 1j * rng.standard_normal(4000)).astype(np.complex64)
    p = mean_input_power(x)
    assert p.dtype == np.float32 and abs(float(p) - float(np.mean(np.abs(x.astype(np.complex128)) ** 2))) < 1e-6 * float(p)


def test_8ms_picks_the_code_that_matches_the_symbol_transition():
    for flip in (False, True):
        x, kw, code, delay = e1_8ms_case(flip)
        o = Galileo8msOracle(**kw)
        assert o.n_bins == 81
        o.set_local_code(code)
        st = o.work(x[:32000])
        assert st == 2 and o.result["code"] == (1 if flip else 0)
        # Acq_delay_samples = indext % samples_per_code (8ms.cc:255): the start of the code inside the block
        assert abs(o.result["acq_delay_samples"] - delay % 16000.0) < 0.5 * 4000000.0 / 1.023e6
        assert abs(o.result["doppler_hz"] - 750.0) < 2.0 / (3 * 8e-3)
        ma, mb = o.rows[o.result["index_doppler"]][0], o.rows[o.result["index_doppler"]][2]
        assert (mb > 4 * ma) if flip else (ma > 4 * mb)
    # noise only, max_dwells 1: negative after one block (8ms.cc:280-287)
    x, kw, code, _ = e1_8ms_case(False, signal=False)
    o = Galileo8msOracle(**kw)
    o.set_local_code(code)
    assert o.work(x[:32000]) == 3


def test_quicksync_known_answer_and_alias_resolution():
    """delay error < 0.5 chip, Doppler error < 2 / (3 * 4 ms) (gps_l1_ca_pcps_quicksync_acquisition_gsoc2014_test.cc:225-228), and the
    direct correlation picks the right one of the folding_factor aliases of the folded delay."""
    for fs, p in ((8000000, 4), (4000000, 4), (8000000, 2)):
        x, kw, code = quicksync_case(fs, p)
        o = QuickSyncOracle(**kw)
        assert o.fft_size == fs // 1000 // p and o.n_bins == 81 and o.n_in == p * fs // 1000
        o.set_local_code(code)
        assert o.work(x) == 2
        spc = fs // 1000
        assert abs(600.0 - o.result["acq_delay_samples"] * 1023.0 / spc) < 0.5
        assert abs(o.result["doppler_hz"] - 750.0) < 2.0 / (3 * 4e-3)
        assert o.result["acq_delay_samples"] == o.result["index_time"] + o.result["alias"] * o.fft_size
        c = np.abs(o.candidates)
        assert c[o.result["alias"]] > 1.5 * np.max(np.delete(c, o.result["alias"]))
    x, kw, code = quicksync_case(signal=False, seed=3)
    o = QuickSyncOracle(**kw)
    o.set_local_code(code)
    assert o.work(x) == 3


def test_fine_doppler_block_as_written_and_with_the_consistent_grid():
    """The block's grid bin i is wiped off at doppler_step * i - doppler_step (fd.cc:170) but reported as i * doppler_step - doppler_max
    (:243).  As written: a 1730 Hz signal is found in bin 4 (1500 Hz), reported as -3000 Hz, and the fine estimate (1737.5 Hz, 12.5 Hz
    resolution) is rejected by the 1 kHz plausibility check (:376).  With the consistent grid the same signal is reported at 1500 Hz and
    refined to 1737.5 Hz."""
    x, kw, code = fine_doppler_case()
    n = 4000
    for consistent in (False, True):
        o = FineDopplerOracle(consistent_grid=consistent, **kw)
        assert o.n_points == 20 and o.fft_size == n
        o.set_local_code(code)
        assert o.dwell(x[:n]) == 1 and o.dwell(x[n:2 * n]) == 2
        assert o.decide() == 3 and float(o.test_statistics) > 5.0
        assert abs(600.0 - o.result["acq_delay_samples"] * 1023.0 / n) < 0.5
        assert o.result["doppler_hz"] == (1500.0 if consistent else -3000.0)
        assert o.estimate_doppler(x[2 * n:]) == 4
        assert abs(o.fine_doppler - 1730.0) <= 12.5
        assert o.result["doppler_hz"] == (o.fine_doppler if consistent else -3000.0)
    x, kw, code = fine_doppler_case(signal=False, seed=5)
    o = FineDopplerOracle(**kw)
    o.set_local_code(code)
    o.dwell(x[:n]); o.dwell(x[n:2 * n])
    assert o.decide() == 5


def test_cccwsr_known_answer_of_the_reference_test():
    """galileo_e1_pcps_cccwsr_ambiguous_acquisition_gsoc2013_test.cc:265-350 + its checks: positive, delay error < 0.5 chip,
    Doppler error < 2 / (3 * 4 ms)."""
    x, kw, cd, cp, delay = cccwsr_case("inphase")
    o = CccwsrOracle(**kw)
    o.set_local_code(cd, cp)
    assert o.work(x[:16000]) == 2
    expected = delay % 16000.0
    err = abs(o.result["acq_delay_samples"] - expected)
    err = min(err, 16000.0 - err)
    assert err < 0.5 * 4000000.0 / 1.023e6
    assert abs(o.result["doppler_hz"] - 750.0) < 2.0 / (3.0 * 4e-3)
    # in-phase components: both branches peak at the same place with nearly equal values (see the oracle's note)
    d = o.result["index_doppler"]
    assert o.rows[d][1] == o.rows[d][3]
    assert abs(o.rows[d][0] - o.rows[d][2]) < 0.2 * o.rows[d][0]


def test_cccwsr_sign_recovery_picks_the_coherent_branch():
    for ds, ps, branch in ((1.0, -1.0, 0), (1.0, 1.0, 1), (-1.0, 1.0, 0), (-1.0, -1.0, 1)):
        x, kw, cd, cp, _ = cccwsr_case("quadrature", data_sign=ds, pilot_sign=ps)
        o = CccwsrOracle(**kw)
        o.set_local_code(cd, cp)
        assert o.work(x[:16000]) == 2
        assert o.result["branch"] == branch, (ds, ps)
        d = o.result["index_doppler"]
        win, lose = (o.rows[d][0], o.rows[d][2]) if branch == 0 else (o.rows[d][2], o.rows[d][0])
        assert win > 8.0 * lose   # the other branch sees data and pilot cancel: only noise is left there


def test_cccwsr_noise_only_negative_and_running_maximum_over_dwells():
    x, kw, cd, cp, _ = cccwsr_case(signal=False)
    o = CccwsrOracle(**kw)
    o.set_local_code(cd, cp)
    assert o.work(x[:16000]) == 3
    # d_mag is cleared in state 0 only (cccwsr.cc:160): dwell 2 on a weaker block keeps dwell 1's peak and its parameters,
    # while the input power is the second block's (cccwsr.cc:192-194)
    xs, kw, cd, cp, _ = cccwsr_case("inphase", n_blocks=1)
    kw2 = dict(kw, max_dwells=2, threshold=1e9)
    o = CccwsrOracle(**kw2)
    o.set_local_code(cd, cp)
    assert o.work(xs[:16000]) == 1
    mag1, res1 = o.mag, dict(o.result)
    assert o.work(x[:16000]) == 3
    assert o.mag == mag1 and o.result == res1
    assert o.input_power == mean_input_power(x[:16000])
    assert o.test_statistics == np.float32(mag1 / mean_input_power(x[:16000]))


def test_cccwsr_branches_are_correlations_with_combined_codes():
    """the identity the product core rests on: data_corr + j pilot_corr == correlation with (data - j pilot), and the
    minus branch with (data + j pilot) -- checked in float64, independently of the oracle and of the GPU path."""
    x, kw, cd, cp, _ = cccwsr_case("quadrature", data_sign=1.0, pilot_sign=1.0)
    o = CccwsrOracle(**kw)
    o.set_local_code(cd, cp)
    o.work(x[:16000])
    d = o.result["index_doppler"]
    n = 16000
    xw = x[:n].astype(np.complex128) * np.exp(-2j * np.pi * (-10000 + 250 * d) / 4000000.0 * np.arange(n))
    X = np.fft.fft(xw)
    for k, code in ((0, cd.astype(np.complex128) - 1j * cp.astype(np.complex128)), (2, cd.astype(np.complex128) + 1j * cp.astype(np.complex128))):
        y = np.fft.ifft(X * np.conj(np.fft.fft(code))) * n
        mag = np.abs(y) ** 2 / float(n) ** 4                       # unnormalised inverse, / fft_size^4 (cccwsr.cc:178, :248)
        t = int(np.argmax(mag))
        if o.rows[d][k] > 8.0 * min(o.rows[d][0], o.rows[d][2]):   # the coherent branch: a defined peak
            assert t == o.rows[d][k + 1]
        assert abs(mag[o.rows[d][k + 1]] - o.rows[d][k]) <= 2e-4 * max(o.rows[d][0], o.rows[d][2])
