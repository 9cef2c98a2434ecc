"""Acquisition batch time at transform lengths beyond one compute unit's plan: split plans (N = S * M, csrc/pcps_onchip.hip) against the
four-step kernels (csrc/pcps_fft.hip), 32 PRN x 41 bins unless noted, CFAR statistic, no grid kept.  Run from the repo root on the GPU."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
CASES = ((25000, 32, False), (50000, 32, False), (50000, 32, True), (32000, 32, False), (32736, 32, False), (40000, 32, False), (65536, 32, False),
         (64000, 32, False), (80000, 8, False), (100000, 32, False), (128000, 32, False), (200000, 16, False))
for n, P, bt in CASES:
    fs = n * 1000
    x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
    code = (np.random.randn(n // 2 if bt else n) + 1j * np.random.randn(n // 2 if bt else n)).astype(np.complex64)
    line = "N %6d%s, %2d PRN x 41 bins:" % (n, " bit-transition" if bt else "", P)
    for name, path in (("on-chip", 0), ("four-step", 1)):
        if n == 25000 and path == 1:
            continue
        acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, consumed_samples=n, bit_transition_flag=bt, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=2,
                                  samples_per_code=float(n), max_prn=P, device=0, keep_grid=False, transform_path=path)
        for p in range(P):
            acq.set_local_code(p, code)
        acq.time_dwells(x, P, reps=10)
        ms = min(acq.time_dwells(x, P, reps=10) for _ in range(3))
        msp = min(acq.time_dwells(x, P, reps=20, pipelined=True) for _ in range(2)) if path == 0 else float("nan")
        alg = 16.0 * n * 41 * (P + 1)
        line += "  %s %.3f ms (pipelined %.3f) = %.0f dwells/s, %.2f TB/s alg;" % (name, ms, msp, P / min(ms, msp if msp == msp else ms) * 1e3, alg / ms / 1e9)
        acq.close()
    print(line, flush=True)
