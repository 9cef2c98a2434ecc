import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, time
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking import CorrelatorBank
import bench
fs, n, C, E, T = 25e6, 25000, 32, 400, 3
dev = torch.device("cuda", 0)
n_samples = (E + 2) * n
x = torch.view_as_complex(torch.randn(n_samples, 2, device=dev).contiguous())
bank = CorrelatorBank(C, 1023, device=0)
for c in range(C):
    bank.set_code(c, oracle.ca_code(c % 32 + 1))
rng = np.random.default_rng(3)
jobs, rows = bench.build_jobs(C, E, n, fs, T, rng.uniform(-5000, 5000, 8), rng.uniform(0, 1023, 8), 0)
bank.upload_jobs(jobs); bank.set_splits(1)
bank.set_stream_device(x.data_ptr(), n_samples, keepalive=x)
torch.cuda.synchronize(); time.sleep(1.0)
print("after 1 s idle:", ["%.1f" % (bank.time_launches(20) * 1e3) for _ in range(8)])
print("long runs (400 launches each):", ["%.1f" % (bank.time_launches(400) * 1e3) for _ in range(6)])
time.sleep(0.5)
print("after 0.5 s idle:", ["%.1f" % (bank.time_launches(5) * 1e3) for _ in range(8)])
