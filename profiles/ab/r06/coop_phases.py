"""Where a period of the cooperating closed loop goes (profiling build: GSH_LIB_PATH=build/variants/lib_coopprof.so, -DGSH_COOP_PROFILE): channel 0's wall-clock
stamps (100 MHz) averaged over the periods of one launch -- published, helper 1 has seen the window, helper 1 has stored its sums, the main work-group's own segment
done, sums gathered."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnss_sdr_amd, oracle
from gnss_sdr_amd import _lib
from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
fs, n, E = 25e6, 25000, 200
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn((E + 3) * n, 2, device=dev).contiguous())
L = _lib.load()
L.gsh_debug_coop_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for ch in (32, 1):
    for G in (2, 4):
        loop = TrackingLoop(trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, enable_lock_detectors=1, max_code_lock_fail=1 << 30, max_carrier_lock_fail=1 << 30), ch, 1023, device=0)
        loop.set_stream_device(x.data_ptr(), x.numel(), keepalive=x)
        loop.set_split(G)
        rng = np.random.default_rng(1)
        for c in range(ch):
            loop.start(c, oracle.ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
        out = (C.c_ulonglong * 8)()
        loop.run(100)
        L.gsh_debug_coop_profile(loop._h, out)   # discard the first launch
        ms = loop.time_run(E - 8, reps=1)
        L.gsh_debug_coop_profile(loop._h, out)
        # time_run: warm-up + 1 rep = two launches' worth of stamps
        v = [int(o) for o in out]
        nh, nm = max(v[3], 1), max(v[7], 1)
        pub, seen, stored, own, gath = v[4] / nm, v[1] / nh, v[2] / nh, v[5] / nm, v[6] / nm
        print("%2d channels, %d work-groups: %.3f us per period | published -> helper saw it %.2f us -> helper's sums stored %.2f us -> gathered %.2f us | main's own segment done %.2f us after publishing, then waits %.2f us"
              % (ch, G, ms * 1e3 / (E - 8), (seen - pub) / 100, (stored - seen) / 100, (gath - stored) / 100, (own - pub) / 100, (gath - own) / 100), flush=True)
        loop.close()
