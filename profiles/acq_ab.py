"""Acquisition A/B timing: 32 PRN x 41 bins dwell batches at a list of lengths (on-chip plans and split plans), single-stream latency and pipelined
throughput (gsh_acq_time_dwells / _pipelined), plus a parity spot check of the peak against numpy for the first length.
  GSH_LIB_PATH=build/variants/lib_<tag>.so python profiles/acq_ab.py [lengths ...]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch

from gnss_sdr_amd.acquisition import PcpsAcquisitionBank

lengths = [int(v) for v in sys.argv[1:]] or [4000, 8000, 16000, 25000, 32768, 50000, 100000, 128000]
dev = torch.device("cuda", 0)
tag = os.path.basename(os.environ.get("GSH_LIB_PATH", "default"))
for n in lengths:
    fs = n * 1000
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, generator=g).contiguous())
    try:
        acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs / 1.023e6)),
                                  samples_per_code=float(n), max_prn=32, device=0, keep_grid=False)
    except Exception as e:
        print(tag, n, "unsupported", e)
        continue
    rng = np.random.default_rng(n)
    code = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for p in range(32):
        acq.set_local_code(p, code)
    acq.time_dwells(x, 32, reps=60, pipelined=True)  # clocks settle
    ms1 = min(acq.time_dwells(x, 32, reps=20) for _ in range(3))
    ms2 = min(acq.time_dwells(x, 32, reps=60, pipelined=True) for _ in range(3))
    res = acq.dwell(x.cpu().numpy(), 32)[0]
    print(f"{tag} N={n:6d} single {ms1 * 1e3:8.1f} us  pipelined {ms2 * 1e3:8.1f} us  dwells/s {32 / (ms2 * 1e-3):9.0f}  peak (tau {res['index_time']}, bin {res['index_doppler']}) stat {res['test_statistics']:.4f}", flush=True)
    acq.close()
