"""CPU tests of the Tong and Galileo 8 ms detector oracles (oracle/pcps_oracle.py: TongOracle, Galileo8msOracle) on the
reference's own synthetic test cases (gps_l1_ca_pcps_tong_acquisition_gsoc2013_test.cc,
galileo_e1_pcps_8ms_ambiguous_acquisition_gsoc2013_test.cc): delay error < 0.5 chip, Doppler error < 2 / (3 T)."""
import numpy as np

from oracle.pcps_oracle import FineDopplerOracle, Galileo8msOracle, QuickSyncOracle, TongOracle, count_doppler_bins, mean_input_power
from detector_cases import e1_8ms_case, fine_doppler_case, quicksync_case, tong_case


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
