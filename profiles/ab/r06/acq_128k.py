"""The 128 000-point split-plan batch (32 PRNs x 41 bins, N = 5 x 25 600) alone and pipelined: one line (A/B of GSH_OC_COMBINE_PARTS / GSH_OC_COMBINE_THREADS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
shapes = ((128000, 32e6),) if len(sys.argv) < 2 else tuple((int(a.split(":")[0]), float(a.split(":")[1])) for a in sys.argv[1:])
for n, fs in shapes:
    x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
    rng = np.random.default_rng(4)
    P = 32
    acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs / 1.023e6)),
                              samples_per_code=float(n), max_prn=P, device=0, keep_grid=False)
    for p in range(P):
        acq.set_local_code(p, (rng.integers(0, 2, n) * 2 - 1).astype(np.complex64))
    acq.time_dwells(x, P, reps=10)
    ms1 = min(acq.time_dwells(x, P, reps=10) for _ in range(3))
    ms2 = min(acq.time_dwells(x, P, reps=20, pipelined=True) for _ in range(3))
    print("N = %6d parts %s threads %s: %.3f ms per batch alone, %.3f pipelined" % (n, os.environ.get("GSH_OC_COMBINE_PARTS", "-"), os.environ.get("GSH_OC_COMBINE_THREADS", "-"), ms1, ms2), flush=True)
    acq.close()
