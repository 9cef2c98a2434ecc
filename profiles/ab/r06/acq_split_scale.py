"""Split-plan acquisition (decimation in time, N = 128 000 = 5 x 25 600; N = 50 000 = 2 x 25 000): time per PRN against the number of PRNs in the batch --
does the Z scratch (n_prn x 41 x N x 8 B) behave differently once it no longer fits the 256 MiB Infinity Cache?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
for n, fs in ((128000, 32e6), (50000, 50e6)):
    x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
    rng = np.random.default_rng(4)
    for P in (1, 2, 4, 8, 16, 32):
        acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs / 1.023e6)),
                                  samples_per_code=float(n), max_prn=P, device=0, keep_grid=False)
        for p in range(P):
            acq.set_local_code(p, (rng.integers(0, 2, n) * 2 - 1).astype(np.complex64))
        acq.time_dwells(x, P, reps=10)
        ms1 = min(acq.time_dwells(x, P, reps=10) for _ in range(3))
        ms2 = min(acq.time_dwells(x, P, reps=20, pipelined=True) for _ in range(3))
        print("N = %6d, %2d PRN: %.3f ms per batch single stream (%.1f us per PRN), %.3f pipelined (%.1f us per PRN); Z scratch %.0f MB" %
              (n, P, ms1, ms1 * 1e3 / P, ms2, ms2 * 1e3 / P, P * 41 * n * 8 / 1e6), flush=True)
        acq.close()
