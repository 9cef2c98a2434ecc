"""On-chip plans against the four-step path for the 2.046 Msps family (GSH_OC_PLANS with the radix-11 / radix-31 butterflies)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
for n in (2046, 4092, 8184, 16368):
    fs = n * 1000
    x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
    code = (np.random.randn(n) + 1j * np.random.randn(n)).astype(np.complex64)
    for path in (0, 1):
        acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=2, samples_per_code=float(n), max_prn=32, device=0, keep_grid=False, transform_path=path)
        for p in range(32):
            acq.set_local_code(p, code)
        acq.time_dwells(x, 32, reps=200)
        ms = min(acq.time_dwells(x, 32, reps=100) for _ in range(3))
        print("N %5d path %d: %.1f us per 32 x 41 batch" % (n, path, ms * 1e3))
        acq.close()
