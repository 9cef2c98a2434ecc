"""GSH_TRK_HOST_RECORDS=0|1 python profiles/ab/r03/loop_host_records.py: wall time of gsh_trk_run (records asked for) at BASELINE config 2's shape -- 32 channels, 25 000
samples per period -- for launches of 1, 5, 20 and 200 periods: kernel + the way the records come back (two copies queued behind the kernel, or the kernel's
own stores into page-locked host memory)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnss_sdr_amd.codes import gps_l1_ca_code
import ctypes

from gnss_sdr_amd._lib import check
from gnss_sdr_amd.tracking_loop import TrackingLoop, TrkEpoch, trk_conf

fs, n, C = 25e6, 25000, 32
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(2 * 1300 * n, generator=g, device="cuda", dtype=torch.float32)
conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0)
loop = TrackingLoop(conf, C, 1023, device=0)
loop.set_stream_device(x.data_ptr(), 1300 * n, keepalive=x)
rng = np.random.default_rng(1)
for c in range(C):
    loop.start(c, gps_l1_ca_code(c + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
loop.time_run(200, reps=20)
out = []
for k in (1, 5, 20, 200):
    reps = 200 // k if k < 200 else 3
    ts = []
    rec = (TrkEpoch * (C * k))()
    done = (ctypes.c_int32 * C)()
    for trial in range(3):
        for c in range(C):
            loop.start(c, gps_l1_ca_code(c + 1), 100 + c, 0, 100.0 * c)
        t0 = time.perf_counter()
        for r in range(reps):
            check(loop._lib.gsh_trk_run(loop._h, k, rec, done))
        ts.append((time.perf_counter() - t0) / reps)
    out.append("%d periods: %.1f us per launch (%.2f us per period)" % (k, min(ts) * 1e6, min(ts) * 1e6 / k))
print("GSH_TRK_HOST_RECORDS=%s  " % os.environ.get("GSH_TRK_HOST_RECORDS", "unset") + "; ".join(out))
loop.close()
