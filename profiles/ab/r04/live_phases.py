"""Phase clocks of the LIVE form of the closed-loop kernel (library built with -DGSH_TRK_PROFILE=1 / 2: GSH_LIB_PATH; GSH_PHASE_DETAIL says which): one residency over a
ring that already holds the block, records read from the host ring afterwards; the launched form over the same stream beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
from gnss_sdr_amd.sample_stream import SampleStream
fs, n, E, ch = 25e6, 25000, 200, 32
x = torch.view_as_complex(torch.randn((E + 3) * n, 2, device="cuda")).contiguous()
conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30)
detail = os.environ.get("GSH_PHASE_DETAIL", "1")
def show(tag, rec):
    f = lambda get: np.array([[get(r) for r in rr[5:E - 5]] for rr in rec]).mean()
    if detail == "2":
        print(tag, "correlation: set-up %.0f  trips %.0f  wave sums %.0f  barrier %.0f  | whole correlation %.0f  serial %.0f" % (
            f(lambda r: r.corr[8]), f(lambda r: r.corr[9]), f(lambda r: r.accu[6]), f(lambda r: r.accu[7]), f(lambda r: r.corr[6]), f(lambda r: r.corr[7])))
    else:
        print(tag, "serial: lanes + meeting %.0f  join %.0f  update_tracking_vars %.0f  symbol+record %.0f  next window (+ live_advance) %.0f | whole correlation %.0f  serial %.0f" % (
            f(lambda r: r.corr[8]), f(lambda r: r.accu[6]), f(lambda r: r.accu[7]), f(lambda r: r.accu[8]), f(lambda r: r.accu[9]) - f(lambda r: r.corr[7]), f(lambda r: r.corr[6]), f(lambda r: r.corr[7])))
def start_all(loop):
    rng = np.random.default_rng(1)
    for c in range(ch):
        loop.start(c, oracle.ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
loop = TrackingLoop(conf, ch, 1023, device=0)
loop.set_stream_device(x.data_ptr(), x.numel(), keepalive=x)
start_all(loop)
rec, done = loop.run(E)
show("launched", rec)
loop.close()
ring = SampleStream(x.numel() + 2 * n, 2 * n, device=0)
ring.push_device(x.data_ptr(), x.numel())
live = TrackingLoop(conf, ch, 1023, device=0)
live.set_stream_ring(ring)
start_all(live)
live.live_configure(idle_timeout_us=2000, residency_us=2000000)
live.live_begin()
t_end = time.time() + 5.0
got = [[] for _ in range(ch)]
while time.time() < t_end and min(len(g) for g in got) < E:
    for c in range(ch):
        r, pending, nw, active = live.live_take(c, 64)
        got[c] += r
live.live_quiesce()
show("live    ", got)
live.close(); ring.close()
