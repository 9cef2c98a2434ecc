"""us per period of one live residency (32 channels, lock detectors on) as bench.py measures it, for the library GSH_LIB_PATH names"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
fs, n, E = 25e6, 25000, 200
dev = torch.device("cuda", 0)
x, dop, cph = bench.make_stream_torch(torch, dev, (E + 3) * n, fs)
for rep in range(3):
    r = bench.closed_loop_metric(0, x, x.numel(), fs, n, dop, cph, channels=32, epochs=E, lock_detectors=True, live=True)
    print(os.path.basename(os.environ.get("GSH_LIB_PATH", "current")), "launched %.3f us  live %.3f us per period" % (r["us_per_epoch"], r["live"].get("us_per_epoch", float("nan"))))
