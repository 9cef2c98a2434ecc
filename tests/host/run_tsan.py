#!/usr/bin/env python3
"""The fake-engine program of tests/test_tracking_adapters.py (adapters + block + Hip_Tracking_Runtime + Hip_Sample_Ring in front of tests/host/fake_gsh_engine.cc)
rebuilt with -fsanitize=thread and run: thirteen signals, restarts, the dump / TOW cases and 32 block threads on one shared runtime, with one device handle for all
of them and with four channels per handle (eight groups).  Needs /root/reference (the adapters compile against its headers); about three minutes.

Then tests/host/test_channel.cc the same way: the reference's own Channel / ChannelFsm / channel_msg_receiver_cc over the HIP adapters -- a channel's whole life, twelve
channels with four of them churning, seven kinds of injected engine failure (the reference's own lock-order inversion is suppressed: tests/host/tsan_reference.supp).

    python tests/host/run_tsan.py            -> exit status 0 and "0 ThreadSanitizer reports" three times
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = "/tmp/test_tracking_adapters_tsan"
real, cmds = subprocess.run, []


def record(cmd, **kw):
    cmds.append(list(cmd))
    return real(cmd, **kw)


subprocess.run = record
g.build_tracking_adapter_test()       # (rebuilds the regular programs too; their command lines are what is wanted)
subprocess.run = real
fake = [c for c in cmds if any(str(x).endswith("_fake") for x in c)]
if not fake:
    sys.exit("the fake-engine program was not built (no /root/reference here?)")
cmd = fake[0]
cmd[cmd.index("-o") + 1] = out
real(cmd[:1] + ["-fsanitize=thread", "-g"] + cmd[1:], check=True)
status = 0
for per_handle in ("", "4"):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0")
    if per_handle:
        env["GSH_TEST_CHANNELS_PER_LAUNCH"] = per_handle
    r = real([out], capture_output=True, text=True, cwd="/tmp", env=env, timeout=1800)
    reports = (r.stdout + r.stderr).count("WARNING: ThreadSanitizer")
    ok = r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout
    print(f"channels per handle {per_handle or 'default'}: {'ok' if ok else 'FAILED'}, {reports} ThreadSanitizer reports")
    status |= 0 if (ok and reports == 0) else 1
# ---- the reference's Channel over the HIP adapters (tests/host/test_channel.cc): channel life, churn and the injected engine failures under ThreadSanitizer
chan = g.build_channel_test(tsan=True)
if not chan:
    sys.exit("tests/host/test_channel_fake_tsan was not built (no /root/reference here?)")
env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 suppressions=" + os.path.join(ROOT, "tests", "host", "tsan_reference.supp"))
r = real([chan, "all"], capture_output=True, text=True, cwd="/tmp", env=env, timeout=1800)
reports = (r.stdout + r.stderr).count("WARNING: ThreadSanitizer")
ok = r.returncode == 0 and "CHANNEL OK" in r.stdout
print(f"channel life / churn / faults: {'ok' if ok else 'FAILED'}, {reports} ThreadSanitizer reports")
if not ok or reports:
    print((r.stdout + r.stderr)[-3000:])
status |= 0 if (ok and reports == 0) else 1
sys.exit(status)
