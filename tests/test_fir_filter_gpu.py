"""GPU tests of the device FIR / frequency-translating filter (gsh_fir_*) against oracle/fir_oracle.py (float64; parity unpinned, see there).
Bar: |gpu - f64| <= 2e-6 * sum_k|h[k]| * max|x| per output (float32 MACs over K taps), and block-partition independence to the bit."""
import numpy as np
import pytest

from oracle.fir_oracle import freq_xlating_fir

pytestmark = pytest.mark.gpu


def _run(gpu, torch, x_np, taps, D, fc, fs, kind, blocks):
    from gnss_sdr_amd.sample_stream import FirFilter
    dev = torch.device("cuda", gpu)
    f = FirFilter(taps, D, fc, fs, kind, device=gpu)
    d_x = torch.from_numpy(x_np).to(dev)
    item = d_x.element_size()
    n = len(x_np)
    cap = n // D + 2
    d_y = torch.zeros(cap, dtype=torch.complex64, device=dev)
    pos, out = 0, 0
    for b in blocks:
        m = min(b, n - pos)
        if m <= 0:
            break
        out += f.process_device(d_x.data_ptr() + item * pos, m, d_y.data_ptr() + 8 * out, cap - out)
        pos += m
    if pos < n:
        out += f.process_device(d_x.data_ptr() + item * pos, n - pos, d_y.data_ptr() + 8 * out, cap - out)
    torch.cuda.synchronize()
    f.close()
    return d_y.cpu().numpy()[:out]


@pytest.mark.parametrize("K,D,fc", [(5, 1, 0.0), (65, 1, 0.0), (33, 2, 1.2e6), (128, 5, -3.3e6), (257, 8, 4.0e6), (1024, 16, 250e3), (6, 1, 2e6)])
def test_complex_input_matches_float64(gpu, K, D, fc):
    torch = pytest.importorskip("torch")
    fs = 25e6
    rng = np.random.default_rng(K)
    n = 200003
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    taps = (np.hamming(K) * np.sinc((np.arange(K) - (K - 1) / 2) / (2.5 * D))).astype(np.float32)
    taps /= taps.sum()
    exp = freq_xlating_fir(x, taps, D, fc, fs)
    one = _run(gpu, torch, x, taps, D, fc, fs, "gr_complex", [n])
    assert len(one) == len(exp) == (n + D - 1) // D
    tol = 2e-6 * np.sum(np.abs(taps)) * np.max(np.abs(x)) * np.sqrt(K)
    assert np.max(np.abs(one - exp)) <= tol, (np.max(np.abs(one - exp)), tol)
    # fed in ragged blocks (shorter than the filter, not multiples of D): bit-identical outputs -- the handle carries history and phase
    ragged = _run(gpu, torch, x, taps, D, fc, fs, "gr_complex", [1, 2, K // 2 + 1, 7, 4099, 65536, 3, 100001])
    assert len(ragged) == len(one) and np.array_equal(ragged.view(np.uint32), one.view(np.uint32))


@pytest.mark.parametrize("kind,dtype", [("float", np.float32), ("short", np.int16), ("byte", np.int8)])
def test_real_input_kinds(gpu, kind, dtype):
    """fcf / scf: real samples at an intermediate frequency -> complex baseband"""
    torch = pytest.importorskip("torch")
    fs, fc, D, K = 16e6, 4e6, 4, 64
    rng = np.random.default_rng(5)
    n = 100000
    t = np.arange(n)
    sig = 20.0 * np.cos(2 * np.pi * (fc + 150e3) / fs * t) + 5.0 * rng.standard_normal(n)
    x = sig.astype(dtype) if dtype == np.float32 else np.clip(np.round(sig), -127, 127).astype(dtype)
    taps = (np.hamming(K) * np.sinc((np.arange(K) - (K - 1) / 2) / (2.2 * D))).astype(np.float32)
    taps /= taps.sum()
    exp = freq_xlating_fir(x.astype(np.float64), taps, D, fc, fs)
    got = _run(gpu, torch, x, taps, D, fc, fs, kind, [12345, 1, 50000])
    assert len(got) == len(exp)
    assert np.max(np.abs(got - exp)) <= 2e-5 * np.max(np.abs(exp)) + 1e-4
    # the tone came down to +150 kHz: phase advances by 2 pi 150e3 D / fs per output
    ph = np.angle(got[200:] * np.conj(got[199:-1]))
    assert abs(np.median(ph) - 2 * np.pi * 150e3 * D / fs) < 0.02


def test_acquisition_through_the_flowgraph_decimator(gpu):
    """GNSS-SDR.use_acquisition_resampler (gnss_flowgraph.cc:1116-1211): 16 Msps stream -> low-pass + decimate by 8 on the GPU ->
    PCPS acquisition at 2 Msps; the code start maps back to the input rate as update_synchro does with the resampler fields
    (acq.cc:584-589): delay * resampler_ratio - resampler_latency_samples."""
    import torch
    import oracle
    from helpers import synth_gps_l1_stream
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    from gnss_sdr_amd.sample_stream import FirFilter, acquisition_resampler_design
    fs, prn, fd, code_phase = 16000000, 7, 1250.0, 311.25
    dec, dec_fs, taps, latency = acquisition_resampler_design(fs, 2e6)
    assert (dec, dec_fs, len(taps), latency) == (8, 2e6, 39, 19)
    n_in = 3 * 16000
    x = synth_gps_l1_stream(n_in, fs, [prn], [fd], [code_phase], cn0_dbhz=50.0, seed_noise=4)
    dev = torch.device("cuda", gpu)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.zeros(n_in // dec + 2, dtype=torch.complex64, device=dev)
    f = FirFilter(taps, dec, 0.0, float(fs), "gr_complex", device=gpu)
    n_out = f.process_device(d_x.data_ptr(), n_in, d_y.data_ptr(), d_y.numel())
    assert n_out == n_in // dec
    f.close()
    n = 2000
    acq = PcpsAcquisitionBank(int(dec_fs), n, 5000, 250, 2, float(n), device=gpu)
    acq.set_local_code(0, oracle.ca_code_complex_sampled(prn, int(dec_fs)))
    first = 1000                                      # any block start: the search is circular in the code period
    r = acq.dwell_device(d_y.data_ptr() + 8 * first, 1)[0]
    acq.close()
    assert abs(r["doppler_hz"] - fd) <= 250 and r["test_statistics"] > 10.0
    # code start in input samples: block start + (delay at the decimated rate) * ratio - filter latency  (acq.cc:584-589)
    start_in = first * dec + r["acq_delay_samples"] * dec - latency
    samples_per_chip = fs / 1.023e6
    true_start = ((1023.0 - code_phase) * samples_per_chip) % 16000.0
    err = (start_in - true_start + 8000.0) % 16000.0 - 8000.0
    assert abs(err) <= dec                            # one decimated sample
