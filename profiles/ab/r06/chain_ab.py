"""Batched correlator, BASELINE config 2: launch time against the chain length (gsh_bank_set_chain_length; 1 = one job per work-group) + bit identity of the outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking import CorrelatorBank
import bench
fs, n, C, E, T = 25e6, 25000, 32, 400, 3
dev = torch.device("cuda", 0)
n_samples = (E + 2) * n
x = torch.view_as_complex(torch.randn(n_samples, 2, device=dev).contiguous())
rng = np.random.default_rng(3)
jobs, rows = bench.build_jobs(C, E, n, fs, T, rng.uniform(-5000, 5000, 8), rng.uniform(0, 1023, 8), 0)
ref = None
ks = [int(k) for k in sys.argv[1:]] or [1, 2, 4, 5, 8, 10, 16, 0]
for rep in range(2):
    for K in ks:
        bank = CorrelatorBank(C, 1023, device=0)
        for c in range(C):
            bank.set_code(c, oracle.ca_code(c % 32 + 1))
        bank.set_chain_length(K)
        bank.upload_jobs(jobs)
        bank.set_splits(1)
        bank.set_stream_device(x.data_ptr(), n_samples, keepalive=x)
        ms = min(bank.time_launches(20) for _ in range(5))
        out = bank.read_outputs()
        if ref is None:
            ref = out.copy()
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        print("chain %2d: %.1f us  -> %.1f M correlators/s   outputs identical to chain %d: %s" % (K, ms * 1e3, C * E * T / ms / 1e3, ks[0], same), flush=True)
        bank.close()
