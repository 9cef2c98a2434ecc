"""noise only, the CARRIER lock counter is the one that passes its limit (max_carrier_lock_fail 20, code limit out of reach): device loop against the oracle loop"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import synth_gps_l1_stream
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
fs, n, epochs = 4e6, 4000, 1100
x = synth_gps_l1_stream((epochs + 3) * n, fs, [], [], [], seed_noise=21)
for kw in (dict(max_carrier_lock_fail=20, max_code_lock_fail=1 << 30), dict(max_carrier_lock_fail=1 << 30, max_code_lock_fail=20), dict(max_carrier_lock_fail=20, max_code_lock_fail=20),
           dict(max_carrier_lock_fail=20, max_code_lock_fail=20, enable_symbol_sync=1, symbols_per_bit=20)):
    conf = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, pull_in_time_s=0, enable_lock_detectors=1, cn0_min=35, **kw)
    loop = TrackingLoop(trk_conf(**conf), 1, 1023, device=0)
    loop.set_stream_host(x)
    loop.start(0, oracle.ca_code(3), 100, 0, 500.0)
    rec, done = loop.run(epochs)
    ora = oracle.trk_run(oracle.trk_conf(**conf), oracle.ca_code(3), x, 100, 0, 500.0, epochs)
    print(kw, "device: periods", int(done[0]), "lost flag", bool(rec[0][int(done[0]) - 1].flags & 2), "| oracle: periods", len(ora), "lost flag", bool(ora[-1].flags & 2))
    loop.close()
