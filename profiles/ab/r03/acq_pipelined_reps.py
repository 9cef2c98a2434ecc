"""Why bench.py's pipelined 25 000-point batch (0.142 ms) and profiles/acq_ab.py's (0.113 ms) differ: same library, same call
(gsh_acq_time_dwells_pipelined), different repetition counts and inputs.  Prints the per-batch time for a sequence of calls."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch

from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
from gnss_sdr_amd.codes import gps_l1_ca_code_sampled

dev = torch.device("cuda", 0)
n, fs = 25000, 25000000
g = torch.Generator(device=dev)
g.manual_seed(n)
x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g).contiguous())
for codes in ("random complex", "GPS C/A"):
    acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=25, samples_per_code=float(n), max_prn=32,
                              device=0, keep_grid=False)
    rng = np.random.default_rng(n)
    for p in range(32):
        acq.set_local_code(p, (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) if codes.startswith("random") else gps_l1_ca_code_sampled(p + 1, fs))
    for reps in (60, 60, 60, 200, 200, 1000, 1000, 60, 200, 3000, 60):
        ms = acq.time_dwells(x, 32, reps=reps, pipelined=True)
        print(f"{codes:15s} pipelined reps={reps:5d}  {ms * 1e3:7.1f} us per batch", flush=True)
    for reps in (20, 200):
        ms = acq.time_dwells(x, 32, reps=reps)
        print(f"{codes:15s} single    reps={reps:5d}  {ms * 1e3:7.1f} us per batch", flush=True)
    acq.close()
