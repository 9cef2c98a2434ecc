"""Launches of the batched correlator (12 800 jobs of BASELINE config 2's shape) back to back on ONE stream against alternating on TWO streams with a bank each
(two launches in flight: the head of one fills the tail of the other).  Wall time per launch over 400 launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking import CorrelatorBank
import bench
n = 25000
fs, C, E, T = 25e6, 32, 400, 3
dev = torch.device("cuda", 0)
block = (E + 2) * n
NB = 8
x = torch.view_as_complex(torch.randn(NB * block, 2, device=dev).contiguous())
jobs, rows = bench.build_jobs(C, E, n, fs, T, [], [], 1)
banks, streams = [], []
for b in range(3):
    bank = CorrelatorBank(C, 1023, device=0)
    for c in range(C):
        bank.set_code(c, oracle.ca_code(c % 32 + 1))
    bank.upload_jobs(jobs)
    bank.set_splits(1)
    bank.set_stream_device(x.data_ptr(), NB * block, keepalive=x)
    banks.append(bank)
    streams.append(torch.cuda.Stream(device=dev))


def run(n_streams, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(reps):
        b = j % n_streams
        banks[b].set_sample_base((j % NB) * block)
        banks[b].launch(streams[b].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for _ in range(2):
    for ns in (1, 2, 3):
        run(ns, 100)
        print("%d stream(s): %.1f us per launch -> %.1f M correlators/s" % (ns, (us := min(run(ns, 400) for _ in range(3))), C * E * T / us))
