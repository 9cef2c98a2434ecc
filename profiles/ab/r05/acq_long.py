"""A few seconds of pipelined acquisition batches (for sampling the clocks / power from outside, and for the long-run rate)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
n, fs = 25000, 25000000
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=25, samples_per_code=float(n), max_prn=32, device=0, keep_grid=False)
code = (np.random.randn(n) + 1j * np.random.randn(n)).astype(np.complex64)
for p in range(32):
    acq.set_local_code(p, code)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    t0 = time.time()
    ms = acq.time_dwells(x, 32, reps=reps, pipelined=True)
    print("chunk %d: %.1f us per batch (%.2f s)" % (k, ms * 1e3, time.time() - t0), flush=True)
