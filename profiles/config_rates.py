"""Correlator-bank throughput at the shapes of BASELINE configs 4 and 5 (one GPU's share), settled clocks.
Run on the GPU box from the repo root:  python profiles/config_rates.py
Codes are random +-1 sequences of the right lengths (the kernel's cost does not depend on the code values)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

rng = np.random.default_rng(1)
VEML = [-1.0, -0.3, 0.0, 0.3, 1.0]


def nco(fs, fd, f_carrier, chip_rate, spc):
    return dict(rem_carr_phase_rad=float(np.float32(rng.uniform(0, 2 * np.pi))), phase_step_rad=float(np.float32(2 * np.pi * fd / fs)),
                rem_code_phase_chips=float(np.float32(rng.uniform(0, 1) * spc)),
                code_phase_step_chips=float(np.float32(chip_rate * (1 + fd / f_carrier) / fs * spc)))


def measure(codes, rows, n_stream, splits=0, device=0):
    """milliseconds per launch at settled clocks"""
    import torch
    import gnss_sdr_amd
    from gnss_sdr_amd.tracking import CorrelatorBank, make_jobs
    x = torch.view_as_complex(torch.randn(n_stream, 2, device=torch.device("cuda", device)).contiguous())
    bank = CorrelatorBank(len(codes), max(len(c) for c in codes), device=device)
    for i, c in enumerate(codes):
        bank.set_code(i, c)
    bank.upload_jobs(make_jobs(rows))
    bank.set_stream_device(x.data_ptr(), n_stream, keepalive=x)
    bank.set_splits(splits)
    bank.time_launches(200)
    ms = min(bank.time_launches(50) for _ in range(4))
    bank.close()
    return ms


def run(name, codes, rows, n_stream, correlators, samples):
    import torch
    import gnss_sdr_amd
    from gnss_sdr_amd.tracking import CorrelatorBank, make_jobs
    x = torch.view_as_complex(torch.randn(n_stream, 2, device=torch.device("cuda", 0)).contiguous())
    bank = CorrelatorBank(len(codes), max(len(c) for c in codes), device=0)
    for i, c in enumerate(codes):
        bank.set_code(i, c)
    bank.upload_jobs(make_jobs(rows))
    bank.set_stream_device(x.data_ptr(), n_stream, keepalive=x)
    for splits in (0, 1, 2, 4, 8, 16):   # 0 = automatic
        bank.set_splits(splits)   # a long window is cut into `splits` work-groups whose partial sums are added by a second tiny kernel
        bank.time_launches(200)
        ms = min(bank.time_launches(50) for _ in range(4))
        print("%s: %d jobs, splits %2d: %.3f ms per launch -> %.1f M correlators/s, %.2f T channel-samples/s"
              % (name, len(rows), splits, ms, correlators / ms / 1e3, samples / ms / 1e9))
    bank.close()


def code(n):
    return np.sign(rng.standard_normal(n)).astype(np.float32)


def config4(E=16):
    # ---- config 4: Galileo E1, 50 channels, fs 32 Msps, N = 128 000 (4 ms), VE/E/P/L/VL on the pilot + 1-tap data correlator
    fs, n = 32e6, 128000
    codes = [code(8184) for _ in range(100)]
    rows = []
    for e in range(E):
        for ch in range(50):
            off = int(rng.integers(0, n)) + e * n
            p = nco(fs, rng.uniform(-4000, 4000), 1575.42e6, 1.023e6, 2)
            rows.append(dict(sample_offset=off, n_samples=n, code_slot=ch, shifts_chips=VEML, **p))
            rows.append(dict(sample_offset=off, n_samples=n, code_slot=50 + ch, shifts_chips=[0.0], **p))
    return "config 4 (E1, 50 channels x %d periods of 4 ms, 5 + 1 taps)" % E, codes, rows, (E + 1) * n + 64, 50 * 6 * E, 50.0 * 2 * n * E



def config5_share():
    # ---- config 5, one GPU's 32 of the 256 channels: 12 GPS L1 + 12 Galileo E1 + 8 GPS L5 at 50 Msps, 40 ms of two RF streams
    fs = 50e6
    n_l1, n_e1, n_l5, span = 50000, 200000, 50000, 40
    len_a = (span + 5) * 50000
    len_b = (span + 2) * 50000
    codes, rows, corr, samp = [], [], 0, 0.0
    for ch in range(12):
        codes.append(code(1023))
        p = nco(fs, rng.uniform(-5000, 5000), 1575.42e6, 1.023e6, 1)
        o = int(rng.integers(0, n_l1))
        for e in range(span):
            rows.append(dict(sample_offset=o + e * n_l1, n_samples=n_l1, code_slot=len(codes) - 1, shifts_chips=[-0.5, 0.0, 0.5], **p))
        corr += 3 * span
        samp += n_l1 * span
    for ch in range(12):
        codes.append(code(8184)); codes.append(code(8184))
        p = nco(fs, rng.uniform(-5000, 5000), 1575.42e6, 1.023e6, 2)
        o = int(rng.integers(0, n_e1))
        for e in range(span // 4):
            rows.append(dict(sample_offset=o + e * n_e1, n_samples=n_e1, code_slot=len(codes) - 2, shifts_chips=VEML, **p))
            rows.append(dict(sample_offset=o + e * n_e1, n_samples=n_e1, code_slot=len(codes) - 1, shifts_chips=[0.0], **p))
        corr += 6 * (span // 4)
        samp += 2.0 * n_e1 * (span // 4)
    for ch in range(8):
        codes.append(code(10230))
        p = nco(fs, rng.uniform(-5000, 5000), 1176.45e6, 10.23e6, 1)
        o = len_a + int(rng.integers(0, n_l5))
        for e in range(span):
            rows.append(dict(sample_offset=o + e * n_l5, n_samples=n_l5, code_slot=len(codes) - 1, shifts_chips=[-0.5, 0.0, 0.5], **p))
        corr += 3 * span
        samp += n_l5 * span
    # a receiver works through time: jobs that read the same stretch of the streams are issued together (and share an XCD's L2)
    rows.sort(key=lambda r: (r["sample_offset"] % len_a if r["sample_offset"] < len_a else r["sample_offset"] - len_a) // 50000)
    return "config 5, one GPU's share (12 L1 + 12 E1 + 8 L5 channels, 40 ms at 50 Msps)", codes, rows, len_a + len_b, corr, samp


if __name__ == "__main__":
    run(*config4())
    run(*config5_share())
