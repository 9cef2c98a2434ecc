"""Pipelined acquisition batches against the number of batches in flight (GSH_ACQ_LANES, read once per process): argv = N fs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gnss_sdr_amd
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
n, fs = int(sys.argv[1]), float(sys.argv[2]); P = 32
x = torch.view_as_complex(torch.randn(n, 2, device="cuda:0").contiguous())
rng = np.random.default_rng(4)
acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs / 1.023e6)),
                          samples_per_code=float(n), max_prn=P, device=0, keep_grid=False)
for p in range(P):
    acq.set_local_code(p, (rng.integers(0, 2, n) * 2 - 1).astype(np.complex64))
acq.time_dwells(x, P, reps=40, pipelined=True)
ms = min(acq.time_dwells(x, P, reps=60, pipelined=True) for _ in range(3))
print("N = %d, lanes %s: %.4f ms per batch pipelined" % (n, os.environ.get("GSH_ACQ_LANES", "2"), ms))
