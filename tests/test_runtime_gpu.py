"""Runs the threaded C++ test of the receiver-side batching runtime (tests/host/test_runtime.cc) on the GPU box: a producer
thread feeds an 8-bit front-end stream into the device ring while 32 channel threads correlate through
Hip_Multicorrelator_Batched; every result is checked against the float64 oracle inside the program."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_runtime")


@pytest.mark.gpu
def test_batching_runtime_threads(gpu):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_host_test()
    r = subprocess.run([BIN, "32", "120"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RUNTIME OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RUNTIME_STATS ")][0]
    st = json.loads(line[len("RUNTIME_STATS "):])
    print(line)
    assert st["avg_batch"] > 4.0          # the rendezvous really batches channels together
    assert st["worst_scale_error"] < 1e-6


@pytest.mark.gpu
def test_batching_runtime_mixed_correlator_flavours(gpu):
    """Odd channels in high-dynamics mode: every rendezvous holds two kernel flavours and is split into one launch per flavour;
    all results (both flavours) are checked against the float64 oracle inside the program."""
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_host_test()
    r = subprocess.run([BIN, "16", "40", "mixed"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RUNTIME OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_batching_runtime_track_pilot_companion(gpu):
    """A track_pilot block calls two correlators per epoch over the same window with the same parameters (trk.cc:1236-1256).  With the
    companion link the pair joins the rendezvous together (one rendezvous per epoch, fused on the device); without it every batch
    only ever sees half of the registered correlators.  Both modes are checked against the float64 oracle inside the program."""
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_host_test()
    rates = {}
    for mode in ("pilot", "pilot_nocompanion"):
        r = subprocess.run([BIN, "16", "60", mode], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "RUNTIME OK" in r.stdout, mode + "\n" + r.stdout[-4000:] + r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RUNTIME_STATS ")][0]
        st = json.loads(line[len("RUNTIME_STATS "):])
        print(mode, line)
        rates[mode] = st
    assert rates["pilot"]["batches"] < rates["pilot_nocompanion"]["batches"]
