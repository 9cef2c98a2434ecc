import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
from helpers import synth_gps_l1_stream
for fs, n in ((25e6, 25000), (4e6, 4000)):
    x = synth_gps_l1_stream(12 * n, fs, [3], [1200.0], [417.3], cn0_dbhz=47.0, seed_noise=31)
    for start in (1000, 1001):
        out = {}
        for G in (1, 2, 4):
            loop = TrackingLoop(trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0), 1, 1023, device=0)
            loop.set_stream_host(x)
            loop.set_split(G)
            loop.start(0, oracle.ca_code(3), start, 0, 1190.0)
            rec, done = loop.run(2)
            out[G] = np.array(list(rec[0][0].corr)[:6])
            loop.close()
        ora = oracle.trk_run(oracle.trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0), oracle.ca_code(3), x, start, 0, 1190.0, 2)
        o = np.array(list(ora[0].corr)[:6])
        print(fs, n, start, "G1-oracle", np.max(np.abs(out[1] - o)), "G2-G1", out[2] - out[1], "G4-G1", out[4] - out[1])
