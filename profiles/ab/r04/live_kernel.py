"""A/B of the closed-loop kernel: launches (gsh_trk_time_run) against ONE live residency over the same resident stream (round 4).
BASELINE config 2 shape: 32 channels, 25 Msps, 25 000-sample periods, E/P/L.  Usage: python profiles/ab/r04/live_kernel.py [channels] [periods]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnss_sdr_amd  # noqa: E402
from gnss_sdr_amd.codes import gps_l1_ca_code  # noqa: E402
from gnss_sdr_amd.sample_stream import SampleStream  # noqa: E402
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf  # noqa: E402
from helpers import synth_gps_l1_stream  # noqa: E402

channels = int(sys.argv[1]) if len(sys.argv) > 1 else 32
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fs, n = 25e6, 25000
total = (periods + 4) * n
rng = np.random.default_rng(3)
prns = list(range(1, 9))
dops = [float(v) for v in rng.uniform(-5000, 5000, 8)]
cphs = [float(v) for v in rng.uniform(0, 1023, 8)]
x = synth_gps_l1_stream(total, fs, prns, dops, cphs, cn0_dbhz=45.0, seed_noise=2)


def start_all(loop):
    r = np.random.default_rng(7)
    for c in range(channels):
        if c < 8:
            f_code = 1.023e6 * (1 + dops[c] / 1575.42e6)
            loop.start(c, gps_l1_ca_code(c + 1), int(round((1023.0 - cphs[c]) / f_code * fs)), 0, dops[c] + float(r.uniform(-20, 20)))
        else:
            loop.start(c, gps_l1_ca_code(c % 32 + 1), int(r.integers(0, n)), 0, float(r.uniform(-5000, 5000)))


for lockdet in (0, 1):
    conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, enable_lock_detectors=lockdet, max_carrier_lock_fail=1000000, max_code_lock_fail=1000000)
    # launches
    loop = TrackingLoop(conf, channels, 1023)
    loop.set_stream_host(x)
    start_all(loop)
    loop.time_run(periods, reps=20)
    ms = loop.time_run(periods, reps=5)
    rec_l, done_l = loop.run(periods)
    loop.close()
    # one residency over a ring that holds the whole stream
    ring = SampleStream(total + 2 * n, 2 * n)
    ring.push(x)
    loop = TrackingLoop(conf, channels, 1023)   # (launches over the ring: the window's address is a 64-bit remainder there)
    loop.set_stream_ring(ring)
    start_all(loop)
    loop.time_run(periods, reps=10)
    ms_ring = loop.time_run(periods, reps=5)
    loop.close()
    live = TrackingLoop(conf, channels, 1023)
    live.set_stream_ring(ring)
    start_all(live)
    live.live_configure(idle_timeout_us=2000, residency_us=2000000)
    got = [[] for _ in range(channels)]
    live.live_begin()
    # the slowest channel's progress between two marks (polling costs a few microseconds per look; the marks are 150 periods apart)
    lo, hi = 30, periods - 20
    t_lo = t_hi = None
    while t_hi is None:
        p = min(live.live_take(c, 0)[1] for c in range(channels))
        now = time.perf_counter()
        if t_lo is None and p >= lo:
            t_lo, p_lo = now, p
        if p >= hi:
            t_hi, p_hi = now, p
    us_live = (t_hi - t_lo) * 1e6 / (p_hi - p_lo)
    while live.live_in_flight():
        pass
    for c in range(channels):
        while True:
            r, pending, nw, act = live.live_take(c, 256)
            got[c] += r
            if not r:
                break
    same = all(b"".join(bytes(memoryview(r)) for r in got[c][:periods]) == b"".join(bytes(memoryview(r)) for r in rec_l[c][:done_l[c]]) for c in range(channels))
    n_live = min(len(g) for g in got)
    print(f"lock detectors {lockdet}: launches {ms * 1e3 / periods:.3f} us per period flat, {ms_ring * 1e3 / periods:.3f} over the ring ({channels} channels x {periods}); "
          f"live residency {us_live:.3f} us per period (slowest channel, periods {p_lo}..{p_hi} of {n_live}); "
          f"records identical to the launched run: {same}")
    live.close()
    ring.close()
