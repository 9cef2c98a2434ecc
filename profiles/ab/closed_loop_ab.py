"""A/B timing of the closed-loop kernel between two builds of the library on the same box (GSH_LIB_PATH selects the build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
fs, n, E = 25e6, 25000, 200
dev = torch.device("cuda", 0)
x = torch.randn((E + 3) * n, 2, device=dev)
x = torch.view_as_complex(x).contiguous()
# GSH_LOOP_AB_CONF: "lock" = with the lock detectors / C/N0 estimator (what the tracking adapters run with); "sync" = + symbol synchronisation (GPS L1 C/A preamble search)
extra = {"": {}, "lock": dict(enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30),
         "sync": dict(enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30, enable_symbol_sync=1, symbols_per_bit=20, pull_in_time_s=0)}[os.environ.get("GSH_LOOP_AB_CONF", "")]
for ch in ([int(os.environ['GSH_LOOP_AB_CH'])] if os.environ.get('GSH_LOOP_AB_CH') else [32, 256]):  # GSH_LOOP_AB_CH: one channel count only (counter passes)
    loop = TrackingLoop(trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, **extra), ch, 1023, device=0)
    loop.set_stream_device(x.data_ptr(), x.numel(), keepalive=x)
    rng = np.random.default_rng(1)
    for c in range(ch):
        loop.start(c, oracle.ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
    ms = min(loop.time_run(E, reps=5) for _ in range(3))
    print(os.environ.get("GSH_LIB_PATH", "current"), os.environ.get("GSH_LOOP_AB_CONF", ""), "channels", ch, "us/epoch %.3f" % (ms * 1e3 / E))
    loop.close()
