"""Records of the closed-loop kernel for a fixed scenario, written to a file so that two builds of the library can be compared byte for byte
(GSH_LIB_PATH=... python profiles/ab/r03/loop_records.py out.bin): 8 channels on a 4 Msps stream with three GPS L1 C/A signals (noise-only channels
lose lock), lock detectors on, 600 periods in three launches; then the same with the FLL on during pull-in and a short pull-in time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from helpers import synth_gps_l1_stream
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf

fs, n = 4e6, 4000
dop = [1200.0, -2300.0, 600.0]
cph = [100.0, 511.5, 900.25]
x = synth_gps_l1_stream(620 * n, fs, [1, 2, 3], dop, cph, seed_noise=5)
out = open(sys.argv[1], "wb")
for kw in (dict(enable_lock_detectors=1, cn0_min=30, max_code_lock_fail=20),
           dict(enable_lock_detectors=1, enable_fll_pull_in=1, pull_in_time_s=0, fll_bw_hz=10.0, cn0_min=30, max_code_lock_fail=20)):
    conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, **kw)
    loop = TrackingLoop(conf, 8, 1023, device=0)
    loop.set_stream_host(x)
    rng = np.random.default_rng(3)
    for c in range(8):
        if c < 3:
            f_code = 1.023e6 * (1 + dop[c] / 1575.42e6)
            loop.start(c, oracle.ca_code(c + 1), int(round((1023.0 - cph[c]) / f_code * fs)), 0, dop[c] + 7.0)
        else:
            loop.start(c, oracle.ca_code(c + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-3000, 3000)))
    total = 0
    for launch in range(3):
        rec, done = loop.run(200)
        for c in range(8):
            k = int(done[c])
            total += k
            out.write(bytes((type(rec[c][0]) * k)(*rec[c][:k])))
    print(kw, "periods recorded", total)
    loop.close()
out.close()
