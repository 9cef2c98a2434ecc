import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
dev = torch.device("cuda", 0)
for n in (4000, 6250, 8000, 10000, 12500, 16000, 20000, 25000, 32768):
    fs = n * 1000
    x = torch.randn(n, 2, device=dev).view(torch.float32)
    x = torch.view_as_complex(x.reshape(n, 2).contiguous())
    try:
        acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs/1.023e6)),
                                  samples_per_code=float(n), max_prn=32, device=0, keep_grid=False)
    except Exception as e:
        print(n, "unsupported", e); continue
    code = (np.random.randn(n) + 1j*np.random.randn(n)).astype(np.complex64)
    for p in range(32): acq.set_local_code(p, code)
    ms1 = acq.time_dwells(x, 32, reps=20)
    ms2 = acq.time_dwells(x, 32, reps=40, pipelined=True)
    cells = 32*41
    print(f"N={n:6d} single {ms1*1e3:8.1f} us  pipelined {ms2*1e3:8.1f} us  per-cell (pipelined, 256 CUs) {ms2*1e3/cells*256:6.2f} us  ns/point {ms2*1e6/cells*256/n:6.3f}")
    acq.close()
