"""Direct resampler: the closed form the device kernel evaluates against the reference block's stateful loop
(direct_resampler_conditioner_cc.cc:72-129, restated statement by statement in oracle/gnss_oracle_loop.c), and the device kernel itself.
Integer index selection: the bar is bit equality of every output sample."""
import math

import numpy as np
import pytest

import oracle

RATIOS = [(25e6, 4e6), (32e6, 4e6), (50e6, 25e6), (4e6, 4e6), (12.5e6, 2.048e6), (2.048e6, 2.046e6), (5e6, 4999999.0), (3e6, 1e6),
          (4e6, 10e6), (2.046e6, 8.184e6), (1e6, 1.000001e6), (7e6, 2e6)]


def _closed_form_indices(fs_in, fs_out, n_in):
    two32 = 1 << 32
    dec = fs_in >= fs_out
    step = math.floor(4294967296.0 * (fs_out / fs_in if dec else fs_in / fs_out))
    if step >= two32:
        return np.arange(n_in)
    idx = []
    j = 0
    while True:
        i = -((-j * two32) // step) if dec else ((j + 1) * step) >> 32
        if i >= n_in:
            break
        idx.append(i)
        j += 1
    return np.array(idx, dtype=np.int64)


@pytest.mark.parametrize("fs_in,fs_out", RATIOS)
def test_closed_form_equals_reference_loop(fs_in, fs_out):
    rng = np.random.default_rng(int(fs_in + fs_out) % 1000)
    n = 20011
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    idx = _closed_form_indices(fs_in, fs_out, n)
    one_call = oracle.direct_resampler(x, fs_in, fs_out)
    assert np.array_equal(one_call[:len(idx)].view(np.uint32), x[idx][:len(one_call)].view(np.uint32))
    assert abs(len(one_call) - len(idx)) <= 1          # the block stops when its output quota or its input runs out
    # cutting the stream into ragged work() calls changes nothing: the phase accumulator carries over
    ragged = oracle.direct_resampler(x, fs_in, fs_out, call_sizes=[1, 7, 512, 3, 1000])
    m = min(len(ragged), len(one_call))
    assert m >= len(idx) - 1024 and np.array_equal(ragged[:m].view(np.uint32), one_call[:m].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("fs_in,fs_out", RATIOS)
def test_device_resampler_equals_reference_loop(gpu, fs_in, fs_out):
    torch = pytest.importorskip("torch")
    from gnss_sdr_amd.sample_stream import direct_resample_device
    dev = torch.device("cuda", gpu)
    rng = np.random.default_rng(7)
    n = 300007
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    exp = oracle.direct_resampler(x, fs_in, fs_out)
    d_x = torch.from_numpy(x).to(dev)
    cap = int(n * max(1.0, fs_out / fs_in)) + 16
    d_y = torch.zeros(cap, dtype=torch.complex64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    # the stream arrives in ragged blocks; outputs are numbered from the start of the stream
    out0, in0, total_cons = 0, 0, 0
    for blk in (1, 4097, 100000, 33, 150000, n):
        hi = min(n, in0 + blk)
        if hi <= in0:
            break
        # a block starts where the previous one's consumption ended (the reference keeps unconsumed samples in its input buffer)
        n_out, n_cons = direct_resample_device(gpu, d_x.data_ptr() + 8 * in0, in0, hi - in0, fs_in, fs_out, out0, d_y.data_ptr() + 8 * out0, cap - out0, hip_stream=st)
        out0 += n_out
        in0 += n_cons
        if n_cons == 0 and n_out == 0 and hi == n:
            break
    torch.cuda.synchronize()
    got = d_y.cpu().numpy()[:out0]
    m = min(len(got), len(exp))
    assert m >= len(exp) - 1 and m > 0.9 * n * min(1.0, fs_out / fs_in)
    assert np.array_equal(got[:m].view(np.uint32), exp[:m].view(np.uint32))


def test_acquisition_resampler_design_matches_windowed_sinc():
    """gnss_flowgraph.cc:1165-1211: decimation = the largest divisor of fs not above fs / acq_fs; taps = firdes::low_pass(1, fs,
    fs_dec / 2.1, fs_dec / 2) = Hamming-windowed sinc with unit DC gain (compared with scipy.signal.firwin, the same definition);
    latency = (ntaps - 1) / 2."""
    from scipy.signal import firwin
    from gnss_sdr_amd.sample_stream import acquisition_resampler_design, firdes_low_pass
    for fs, acq_fs, want_dec in ((16000000, 2e6, 8), (25000000, 2e6, 10), (4000000, 2e6, 2), (50000000, 10e6, 5), (2000000, 2e6, 1),
                                 (3000000, 2e6, 1), (12500000, 2e6, 5)):
        dec, dec_fs, taps, latency = acquisition_resampler_design(fs, acq_fs)
        assert dec == want_dec and fs % dec == 0 and dec_fs == fs / dec
        if dec == 1:
            assert len(taps) == 0 and latency == 0
            continue
        ntaps = int(53.0 * fs / (22.0 * dec_fs / 2))
        ntaps += 1 - ntaps % 2
        assert len(taps) == ntaps and latency == (ntaps - 1) // 2
        ref = firwin(ntaps, dec_fs / 2.1, window="hamming", fs=fs)
        assert np.max(np.abs(taps - ref)) < 2e-7 and abs(float(np.sum(taps.astype(np.float64))) - 1.0) < 1e-6
        assert np.array_equal(taps, taps[::-1])        # linear phase: the group delay is exactly `latency` input samples
    assert abs(float(np.sum(firdes_low_pass(2.5, 1e6, 1e5, 5e4).astype(np.float64))) - 2.5) < 1e-5
