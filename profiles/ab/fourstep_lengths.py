import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
for n, P in ((50000, 32), (100000, 32), (80000, 8), (32000, 32), (6625, 32)):
    fs = n * 1000 if n in (50000, 6625) else (25000000 if n == 100000 else 4000000)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
    code = (np.random.randn(n) + 1j * np.random.randn(n)).astype(np.complex64)
    acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=2, samples_per_code=float(n), max_prn=P, device=0, keep_grid=False)
    for p in range(P):
        acq.set_local_code(p, code)
    acq.time_dwells(x, P, reps=20)
    ms = min(acq.time_dwells(x, P, reps=10) for _ in range(3))
    alg = 16.0 * n * 41 * (P + 1)
    print("N %6d, %2d PRN x 41 bins (four-step): %.3f ms per batch = %.0f dwells/s, %.2f TB/s algorithmic" % (n, P, ms, P / ms * 1e3, alg / ms / 1e9))
    acq.close()
