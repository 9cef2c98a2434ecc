"""A few launches of the batched correlator over 12 800 jobs of n samples (argv: n [chain length]) -- for counter runs (profiles/ab/r06/pmc_mcorr.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking import CorrelatorBank
import bench
n = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fs, C, E, T = 25e6, 32, 400, 3
dev = torch.device("cuda", 0)
n_samples = (E + 2) * n
x = torch.view_as_complex(torch.randn(n_samples, 2, device=dev).contiguous())
bank = CorrelatorBank(C, 1023, device=0)
for c in range(C):
    bank.set_code(c, oracle.ca_code(c % 32 + 1))
jobs, rows = bench.build_jobs(C, E, n, fs, T, [], [], 1)
if hasattr(bank, "set_chain_length"):
    bank.set_chain_length(K)
bank.upload_jobs(jobs)
bank.set_splits(1)
bank.set_stream_device(x.data_ptr(), n_samples, keepalive=x)
ms = min(bank.time_launches(10) for _ in range(3))
print("n = %d chain %d: %.1f us" % (n, K, ms * 1e3))
