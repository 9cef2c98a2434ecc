"""Phase timing of the closed-loop kernel (library built with -DGSH_TRK_PROFILE: GSH_LIB_PATH): shader clocks spent per period in the correlation
(window load .. barrier) and in thread 0's loop arithmetic, from the records."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
fs, n, E = 25e6, 25000, 200
x = torch.view_as_complex(torch.randn((E + 3) * n, 2, device="cuda")).contiguous()
# GSH_LOOP_AB_CONF=lock: with the lock detectors / C/N0 estimator (what the tracking adapters run with)
extra = dict(enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30) if os.environ.get("GSH_LOOP_AB_CONF", "") == "lock" else {}
for ch in (32,):
    loop = TrackingLoop(trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, **extra), ch, 1023, device=0)
    loop.set_stream_device(x.data_ptr(), x.numel(), keepalive=x)
    rng = np.random.default_rng(1)
    for c in range(ch):
        loop.start(c, oracle.ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
    ms = loop.time_run(E, reps=3)
    rec, done = loop.run(E)
    tc = np.array([[r.corr[6] for r in rr] for rr in rec]); ts = np.array([[r.corr[7] for r in rr] for rr in rec])
    if os.environ.get("GSH_PHASE_DETAIL") == "2":  # library built with -DGSH_TRK_PROFILE=2
        f = lambda get: np.array([[get(r) for r in rr] for rr in rec])[:, 5:].mean()
        print("  correlation phase, clocks: set-up %.0f  trips %.0f  wave sums + partials %.0f  barrier %.0f  sum over waves + barrier %.0f  read-out + barrier %.0f" % (
            f(lambda r: r.corr[8]), f(lambda r: r.corr[9]), f(lambda r: r.accu[6]), f(lambda r: r.accu[7]), f(lambda r: r.accu[8]), f(lambda r: r.accu[9])))
    elif os.environ.get("GSH_PHASE_DETAIL") == "3":  # library built with -DGSH_TRK_PROFILE=3
        f = lambda get: np.array([[get(r) for r in rr] for rr in rec])[:, 5:].mean()
        print("  lanes done, clocks after the correlation: carrier %.0f  code %.0f  C/N0 %.0f  carrier lock + record %.0f   (meeting over at %.0f)" % (
            f(lambda r: r.accu[6]), f(lambda r: r.accu[7]), f(lambda r: r.accu[8]), f(lambda r: r.corr[9]), f(lambda r: r.corr[8])))
    elif os.environ.get("GSH_PHASE_DETAIL"):
        f = lambda get: np.array([[get(r) for r in rr] for rr in rec])[:, 5:].mean()
        print("  serial section, clocks: lanes side by side + meeting %.0f  join %.0f  update_tracking_vars %.0f  symbol+record %.0f  publish %.0f" % (
            f(lambda r: r.corr[8]), f(lambda r: r.accu[6]), f(lambda r: r.accu[7]), f(lambda r: r.accu[8]), f(lambda r: r.accu[9]) - f(lambda r: r.corr[7])))
    print(os.path.basename(os.environ.get("GSH_LIB_PATH", "current")), os.environ.get("GSH_LOOP_AB_CONF", ""), "channels", ch, "us/epoch %.3f" % (ms * 1e3 / E), "correlation clocks avg %.0f  serial clocks avg %.0f  (clock64 units)" % (tc[:, 5:].mean(), ts[:, 5:].mean()))
    loop.close()
