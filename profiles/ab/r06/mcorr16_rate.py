"""The 16-bit correlator family at BASELINE config 2's shape (32 channels x E epochs of 25 000 samples, E/P/L, complex int16): milliseconds per launch of the
two kernels, correlators per second, and the reference's own Cpu_Multicorrelator_16sc timed on one host core beside it (oracle/_ref, when it travelled)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking16 import CorrelatorBank16, make_job16
n, channels = 25000, 32
rng = np.random.default_rng(2)
for epochs in [int(a) for a in (sys.argv[1:] or ["400", "40", "1"])]:
    x = rng.integers(-50, 51, size=((epochs + 1) * n, 2)).astype(np.int16)
    xd = torch.from_numpy(x).to("cuda:0")
    bank = CorrelatorBank16(channels, 1023, device=0)
    for c in range(channels):
        bank.set_code(c, np.stack([oracle.ca_code(c + 1), np.zeros(1023, np.float32)], -1).astype(np.int16))
    bank.set_stream_device(xd.data_ptr(), len(x), keepalive=xd)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    jobs = [make_job16(e * n + int(rng.integers(0, n)), n, c, float(rng.uniform(0, 6.28)), float(2 * np.pi * rng.uniform(-5000, 5000) / 25e6), float(rng.uniform(0, 1)), 1.023e6 / 25e6, shifts)
            for e in range(epochs) for c in range(channels)]
    bank.upload(jobs)
    ms = min(bank.time_launches(5) for _ in range(3))
    print("%6d jobs: %.3f ms per launch = %.2f M correlators/s" % (len(jobs), ms, len(jobs) * 3 / ms / 1e3), flush=True)
    bank.close()
R = oracle.ref()
if R is not None and hasattr(R, "ref_mcorr16_time"):
    x = rng.integers(-50, 51, size=(50 * n, 2)).astype(np.int16)
    code = np.stack([oracle.ca_code(1), np.zeros(1023, np.float32)], -1).astype(np.int16)
    out = np.zeros((3, 2), np.int16)
    for simd in (0, 1):
        R.ref_set_flavour(simd)
        E = 200
        s = R.ref_mcorr16_time(code.reshape(-1), 1023, np.array([-0.5, 0, 0.5], np.float32), 3, x.reshape(-1), len(x), n, E, 0.3, 0.001, 0.2, 1.023e6 / 25e6, out.reshape(-1))
        print("reference Cpu_Multicorrelator_16sc, %s protokernels, one core: %.1f us per call = %.3f M correlators/s" % ("SIMD" if simd else "generic", s / E * 1e6, 3 * E / s / 1e6))
    R.ref_set_flavour(0)
