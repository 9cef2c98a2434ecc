"""One split-plan acquisition batch shape under a profiler: argv = N fs [n_prn]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
n, fs = int(sys.argv[1]), float(sys.argv[2]); P = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
rng = np.random.default_rng(4)
acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs / 1.023e6)),
                          samples_per_code=float(n), max_prn=P, device=0, keep_grid=False)
for p in range(P):
    acq.set_local_code(p, (rng.integers(0, 2, n) * 2 - 1).astype(np.complex64))
acq.time_dwells(x, P, reps=10)
print("ms per batch, single stream:", acq.time_dwells(x, P, reps=20))
