"""Closed loop at BASELINE config 4 (50 Galileo E1 channels, 128 000-sample windows, 5 + 1 taps, detectors on) against the number of cooperating work-groups per
channel (gsh_trk_set_split); bench.py's own leg with the switch."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import gnss_sdr_amd
import bench
for split in (1, 2, 3, 4, 1, 2, 4):
    r = bench.closed_loop_config4_metric(torch, 0, split=split)
    print("config 4, %d work-group(s) per channel: %.2f us per period, locked %s, %.2f M correlators/s" % (split, r["us_per_epoch"], r["channels_with_signal_locked"], r["value"] / 1e6), flush=True)
