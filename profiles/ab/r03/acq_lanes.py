"""Two against three acquisition batches in flight (gsh_acq_time_dwells_pipelined; the third lane is profiles/ab/r03/acq_three_lanes.patch, not in the tree:
GSH_ACQ_PIPELINE_LANES=3 has no effect without it).  Measured at the end of round 3, same box, alternating:
    lanes 2   112.5 112.5 112.4 112.5 / 112.9 112.9 112.9 112.7 us per batch
    lanes 3   111.7 111.3 111.7 111.5 / 111.6 111.5 111.5 111.7 us per batch"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
from gnss_sdr_amd.codes import gps_l1_ca_code_sampled
dev = torch.device("cuda", 0)
n, fs = 25000, 25000000
g = torch.Generator(device=dev); g.manual_seed(n)
x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g).contiguous())
acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=25, samples_per_code=float(n), max_prn=32, device=0, keep_grid=False)
for p in range(32):
    acq.set_local_code(p, gps_l1_ca_code_sampled(p + 1, fs))
acq.time_dwells(x, 32, reps=1500, pipelined=True)
out = [acq.time_dwells(x, 32, reps=600, pipelined=True) * 1e3 for _ in range(4)]
print("lanes", os.environ.get("GSH_ACQ_PIPELINE_LANES", "2"), " ".join("%.1f" % v for v in out), "us per batch")
