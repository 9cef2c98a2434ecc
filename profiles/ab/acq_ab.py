"""A/B timing of the acquisition batch between two builds (GSH_LIB_PATH selects the build)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import gnss_sdr_amd
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
n, fs = 25000, 25000000
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=25, samples_per_code=float(n), max_prn=32, device=0, keep_grid=False)
code = (np.random.randn(n) + 1j * np.random.randn(n)).astype(np.complex64)
for p in range(32):
    acq.set_local_code(p, code)
acq.time_dwells(x, 32, reps=40, pipelined=True)
ms1 = min(acq.time_dwells(x, 32, reps=40) for _ in range(3))
ms2 = min(acq.time_dwells(x, 32, reps=100, pipelined=True) for _ in range(3))
print(os.environ.get("GSH_LIB_PATH", "current"), "single %.1f us  pipelined %.1f us" % (ms1 * 1e3, ms2 * 1e3))
