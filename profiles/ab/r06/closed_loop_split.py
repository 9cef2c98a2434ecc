"""Closed loop, BASELINE config 2 (32 channels, 25 Msps, detectors on): microseconds per period against the number of cooperating work-groups per channel
(gsh_trk_set_split), and how far the records move (the sums are formed in another order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
from helpers import synth_gps_l1_stream
fs, n, E = 25e6, 25000, 200
dev = torch.device("cuda", 0)
prns = list(range(1, 9))
rng = np.random.default_rng(1)
dops = rng.uniform(-5000, 5000, 8)
cphs = rng.uniform(0, 1023, 8)
x = synth_gps_l1_stream((E + 3) * n, fs, prns, list(dops), list(cphs), cn0_dbhz=45.0, seed_noise=2)
xd = torch.from_numpy(x).to(dev)
extra = dict(enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30)
ref = None
for ch in (32,):
    for G in ([1, 2, 4, 8] if ch <= 32 else [1]):
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
        if ref is None:
            ref = rec
        worst = 0.0
        same_windows = True
        for c in range(ch):
            for a, b in zip(rec[c][:done[c]], ref[c]):
                same_windows = same_windows and a.sample_counter == b.sample_counter
                pa, pb = np.array(list(a.corr)[:6]), np.array(list(b.corr)[:6])
                worst = max(worst, float(np.max(np.abs(pa - pb)) / max(np.hypot(pb[2], pb[3]), 50.0)))
        lock = np.mean([rec[c][-1].carrier_lock_test for c in range(8)])
        print("channels %3d, %d work-group(s) per channel: %.3f us per period; periods done %d..%d; windows identical to the one-work-group run: %s; worst correlator difference %.2e of the prompt; lock test %.3f"
              % (ch, G, ms * 1e3 / (E - 8), min(done), max(done), same_windows, worst, lock), flush=True)
        loop.close()
