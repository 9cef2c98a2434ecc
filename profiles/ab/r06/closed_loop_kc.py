"""Closed loop, BASELINE config 2 (32 channels, 25 Msps, detectors on) and config 4's shape: microseconds per period, records hashed (A/B of the look-up addressing:
GSH_LIB_PATH chooses the library)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
from helpers import synth_gps_l1_stream
import ctypes
fs, n, E = 25e6, 25000, 200
dev = torch.device("cuda", 0)
prns = list(range(1, 9))
rng = np.random.default_rng(1)
dops = rng.uniform(-5000, 5000, 8)
cphs = rng.uniform(0, 1023, 8)
x = synth_gps_l1_stream((E + 3) * n, fs, prns, list(dops), list(cphs), cn0_dbhz=45.0, seed_noise=2)
xd = torch.from_numpy(x).to(dev)
extra = dict(enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30)
for G in (1, 2):
    ch = 32
    loop = TrackingLoop(trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, **extra), ch, 1023, device=0)
    loop.set_stream_device(xd.data_ptr(), xd.numel(), keepalive=xd)
    loop.set_split(G)
    for c in range(ch):
        k = c % 8
        f_code = 1.023e6 * (1 + dops[k] / 1575.42e6)
        start = int(round((1023.0 - cphs[k]) / f_code * fs)) + (c // 8) * n
        loop.start(c, oracle.ca_code(prns[k]), start, 0, float(dops[k]) + 10.0)
    ms = min(loop.time_run(E - 8, reps=5) for _ in range(3))
    rec, done = loop.run(E - 8)
    h = hashlib.sha256()
    for c in range(ch):
        for r in rec[c][:done[c]]:
            h.update(bytes(r))
    print("config 2, %d work-group(s) per channel: %.3f us per period; records %s" % (G, ms * 1e3 / (E - 8), h.hexdigest()[:16]))
