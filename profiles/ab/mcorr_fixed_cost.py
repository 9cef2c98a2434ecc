"""Per-job fixed cost of the open-loop correlator kernel: launch time over 12 800 jobs as a function of the window length (T = a + b n)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking import CorrelatorBank
import bench
fs, C, E, T = 25e6, 32, 400, 3
dev = torch.device("cuda", 0)
res = []
for n in (25000, 12500, 5000, 2048, 1024):
    n_samples = (E + 2) * n
    x = torch.view_as_complex(torch.randn(n_samples, 2, device=dev).contiguous())
    bank = CorrelatorBank(C, 1023, device=0)
    for c in range(C):
        bank.set_code(c, oracle.ca_code(c % 32 + 1))
    rng = np.random.default_rng(3)
    jobs, rows = bench.build_jobs(C, E, n, fs, T, [], [], 1)   # no embedded signals: window offsets random inside the first period
    bank.upload_jobs(jobs)
    bank.set_splits(1)
    bank.set_stream_device(x.data_ptr(), n_samples, keepalive=x)
    ms = min(bank.time_launches(20) for _ in range(5))
    res.append((n, ms * 1e3))
    print("n = %6d: %.1f us per launch of %d jobs" % (n, ms * 1e3, C * E), flush=True)
    bank.close()
(n1, t1), (n2, t2) = res[0], res[-1]
b = (t1 - t2) / (n1 - n2)
a = t1 - b * n1
print("T = %.1f us + %.4f us per 1000 samples per job; at n = 25000 the fixed part is %.0f %% of the launch" % (a, b * 1000, 100 * a / t1))
