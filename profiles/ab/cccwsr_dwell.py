"""CCCWSR dwell on the engine: N = 16000 (Galileo E1, 4 ms at 4 Msps), 81 Doppler bins, two code slots per satellite
(data - j pilot, data + j pilot).  One satellite per dwell (what one channel of the block does) and 16 satellites batched."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
n, fs, bins = 16000, 4000000, 81
x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
code = (np.sign(np.random.randn(n)) + 1j * np.sign(np.random.randn(n))).astype(np.complex64)
for path in (0, 1):
    for slots in (2, 32):
        try:
            acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=10000, doppler_step=250, num_doppler_bins=bins, samples_per_chip=1,
                                      samples_per_code=float(n), max_prn=slots, device=0, keep_grid=False, transform_path=path)
        except Exception as e:
            print("N %d path %d slots %d: not available (%s)" % (n, path, slots, str(e)[:80]))
            continue
        for p in range(slots):
            acq.set_local_code(p, code)
        acq.time_dwells(x, slots, reps=300)
        ms = min(acq.time_dwells(x, slots, reps=100) for _ in range(3))
        print("CCCWSR N %d, %d bins, path %d, %2d slots (%2d satellites): %.1f us per dwell = %.1f us per satellite" % (n, bins, path, slots, slots // 2, ms * 1e3, ms * 1e3 / (slots // 2)))
        acq.close()
