"""GPU tests of the 16-bit correlator family (gsh_mcorr16_* / gsh_bank16_*) through the C ABI: BIT-EXACT against the golden vectors minted
from the reference's own Cpu_Multicorrelator_16sc and against the pinned restatement oracle.mcorr16 (integer work: no tolerance anywhere)."""
import numpy as np
import pytest

import oracle
from test_mcorr16_oracle import golden_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,case", list(golden_cases()), ids=[n for n, _ in golden_cases()])
def test_class_surface_reproduces_the_reference_vectors(gpu, name, case):
    """the reference class's call order: init -> set_local_code_and_taps -> set_input_output_vectors -> Carrier_wipeoff_multicorrelator_resampler"""
    from gnss_sdr_amd.tracking16 import HipMulticorrelator16sc
    x, code, shifts = np.ascontiguousarray(case["x"]), np.ascontiguousarray(case["code"]), np.ascontiguousarray(case["shifts"])
    p = [float(v) for v in case["par"]]
    mc = HipMulticorrelator16sc(gpu)
    out = np.full((len(shifts), 2), 77, np.int16)
    assert mc.init(2 * len(x), len(shifts))
    assert mc.set_local_code_and_taps(len(code), code, shifts)
    assert mc.set_input_output_vectors(out, x)
    assert mc.Carrier_wipeoff_multicorrelator_resampler(p[0], p[1], p[2], p[3], len(x))
    assert np.array_equal(out, case["out"]), (name, out.tolist(), case["out"].tolist())
    assert mc.free()
    mc.close()


def test_borrowed_taps_and_windows_change_between_calls(gpu):
    from gnss_sdr_amd.tracking16 import HipMulticorrelator16sc
    rng = np.random.default_rng(3)
    x = rng.integers(-80, 81, size=(9000, 2)).astype(np.int16)
    code = np.stack([oracle.ca_code(21), np.zeros(1023, np.float32)], -1).astype(np.int16)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    mc = HipMulticorrelator16sc(gpu)
    mc.init(8000, 3)
    mc.set_local_code_and_taps(1023, code, shifts)
    out = np.zeros((3, 2), np.int16)
    for e in range(4):
        win = np.ascontiguousarray(x[e * 331: e * 331 + 4000 + e])
        mc.set_input_output_vectors(out, win)
        mc.Carrier_wipeoff_multicorrelator_resampler(0.2 * e, 0.01 * (e - 1.5), 0.25 * e, 0.2557, len(win))
        assert np.array_equal(out, oracle.mcorr16(code, shifts, win, 0.2 * e, 0.01 * (e - 1.5), 0.25 * e, 0.2557)), e
        shifts[0] -= 0.05  # borrowed: the next call sees the moved taps
        shifts[2] += 0.05
    mc.close()


def _random_jobs(rng, n_jobs, stream_len, n_lo, n_hi, n_codes, taps_choices=(1, 3, 5)):
    from gnss_sdr_amd.tracking16 import make_job16
    jobs, plain = [], []
    for _ in range(n_jobs):
        n = int(rng.integers(n_lo, n_hi + 1))
        off = int(rng.integers(0, stream_len - n + 1))
        nt = int(rng.choice(taps_choices))
        shifts = np.sort(rng.uniform(-1.5, 1.5, nt)).astype(np.float32)
        par = (float(np.float32(rng.uniform(0, 6.28))), float(np.float32(rng.uniform(-0.4, 0.4))), float(np.float32(rng.uniform(0, 3))), float(np.float32(rng.uniform(0.02, 0.6))))
        slot = int(rng.integers(0, n_codes))
        jobs.append(make_job16(off, n, slot, par[0], par[1], par[2], par[3], shifts))
        plain.append((off, n, slot, par, shifts))
    return jobs, plain


def _check(out, plain, codes, x):
    for j, (off, n, slot, par, shifts) in enumerate(plain):
        want = oracle.mcorr16(codes[slot], shifts, x[off:off + n], *par)
        assert np.array_equal(out[j, :len(shifts)], want), (j, n, out[j, :len(shifts)].tolist(), want.tolist())
        assert not out[j, len(shifts):].any(), j


@pytest.mark.parametrize("amp", [60, 2000])
def test_bank_of_many_jobs_is_bit_exact(gpu, amp):
    """more jobs than the device has SIMDs: waves of the rotation kernel walk several jobs' phasor chains side by side; windows of every length around the
    tile and wave sizes; mixed tap counts in one launch; two codes (one real +-1, one complex); amp 2000 makes sums saturate and come back"""
    from gnss_sdr_amd.tracking16 import CorrelatorBank16
    rng = np.random.default_rng(amp)
    x = rng.integers(-amp, amp + 1, size=(60000, 2)).astype(np.int16)
    codes = [np.stack([oracle.ca_code(8), np.zeros(1023, np.float32)], -1).astype(np.int16), rng.integers(-5, 6, size=(2046, 2)).astype(np.int16)]
    bank = CorrelatorBank16(2, 2046, device=gpu)
    for i, c in enumerate(codes):
        bank.set_code(i, c)
    bank.set_stream_host(x)
    jobs, plain = _random_jobs(rng, 2600, len(x), 0, 2300, 2)
    out = bank.correlate(jobs)
    _check(out, plain, codes, x)
    # a second, smaller batch on the same handle (one job per wave), then an eight-tap one
    jobs, plain = _random_jobs(rng, 37, len(x), 8000, 9000, 2)
    _check(bank.correlate(jobs), plain, codes, x)
    jobs, plain = _random_jobs(rng, 5, len(x), 1000, 30000, 2, taps_choices=(8,))
    _check(bank.correlate(jobs), plain, codes, x)
    bank.close()


def test_baseline_shape_on_a_device_resident_stream(gpu):
    """BASELINE config 2's shape in 16 bits: 32 channels x 40 epochs of 25 000 samples, E/P/L, one shared stream that already lies in device memory"""
    import torch
    from gnss_sdr_amd.tracking16 import CorrelatorBank16, make_job16
    rng = np.random.default_rng(2)
    n, epochs, channels = 25000, 40, 32
    x = rng.integers(-50, 51, size=((epochs + 1) * n, 2)).astype(np.int16)
    xd = torch.from_numpy(x).to(torch.device("cuda", gpu))
    codes = [np.stack([oracle.ca_code(p + 1), np.zeros(1023, np.float32)], -1).astype(np.int16) for p in range(channels)]
    bank = CorrelatorBank16(channels, 1023, device=gpu)
    for i, c in enumerate(codes):
        bank.set_code(i, c)
    bank.set_stream_device(xd.data_ptr(), len(x), keepalive=xd)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    jobs, plain = [], []
    for e in range(epochs):
        for c in range(channels):
            par = (float(np.float32(rng.uniform(0, 6.28))), float(np.float32(2 * np.pi * rng.uniform(-5000, 5000) / 25e6)), float(np.float32(rng.uniform(0, 1))), float(np.float32(1.023e6 / 25e6)))
            off = e * n + int(rng.integers(0, n))
            jobs.append(make_job16(off, n, c, *par, shifts))
            plain.append((off, n, c, par, shifts))
    out = bank.correlate(jobs)
    _check(out, plain, codes, x)
    ms = bank.time_launches(3)
    assert ms > 0
    bank.close()


def test_errors_are_reported_not_computed(gpu):
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.tracking16 import CorrelatorBank16, HipMulticorrelator16sc, make_job16
    bank = CorrelatorBank16(2, 1023, device=gpu)
    code = np.ones((1023, 2), np.int16)
    with pytest.raises(GshError):
        bank.correlate([make_job16(0, 10, 0, 0, 0, 0, 0.1, [0.0])])  # no stream
    bank.set_stream_host(np.zeros((100, 2), np.int16))
    with pytest.raises(GshError):
        bank.correlate([make_job16(0, 10, 0, 0, 0, 0, 0.1, [0.0])])  # code slot not set
    bank.set_code(0, code)
    with pytest.raises(GshError):
        bank.correlate([make_job16(95, 10, 0, 0, 0, 0, 0.1, [0.0])])  # window leaves the stream
    with pytest.raises(GshError):
        bank.correlate([make_job16(0, 10, 0, float("nan"), 0, 0, 0.1, [0.0])])
    with pytest.raises(GshError):
        bank.set_code(0, np.ones((2000, 2), np.int16))  # longer than the bank was sized for
    out = bank.correlate([make_job16(0, 0, 0, 0, 0, 0, 0.1, [0.0, 0.5])])  # an empty window: zeros, as the reference's loop leaves them
    assert not out.any()
    bank.close()
    mc = HipMulticorrelator16sc(gpu)
    with pytest.raises(GshError):
        mc.Carrier_wipeoff_multicorrelator_resampler(0.0, 0.0, 0.0, 0.1, 10)  # nothing set
    mc.init(100, 3)
    mc.set_local_code_and_taps(1023, code, np.zeros(3, np.float32))
    out = np.zeros((3, 2), np.int16)
    mc.set_input_output_vectors(out, np.zeros((100, 2), np.int16))
    with pytest.raises(GshError):
        mc.Carrier_wipeoff_multicorrelator_resampler(0.0, 0.0, 0.0, 0.1, 101)  # longer than init() sized
    out[:] = 5
    mc.Carrier_wipeoff_multicorrelator_resampler(0.0, 0.0, 0.0, 0.1, 0)  # zero samples: the reference's loop leaves zeros
    assert not out.any()
    mc.close()
