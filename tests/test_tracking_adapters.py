"""tests/host/test_tracking_adapters: the TrackingInterface adapters (GPS_L1_CA_DLL_PLL_Tracking_HIP, Galileo_E1_DLL_PLL_VEML_Tracking_HIP,
GPS_L5_DLL_PLL_Tracking_HIP, Galileo_E5a_DLL_PLL_Tracking_HIP), the dll_pll_veml_tracking_hip GNU Radio block and the Dll_Pll_Conf ->
gsh_trk_conf mapper, compiled against the reference's own headers and checked against the reference's own tracking chain
(oracle/_ref/libgnsssdr_ref_trk.so).  The binary is built by __graft_entry__.build() where /root/reference is present and travels with the tree.

CPU: `conf` mode -- the mapper equals what dll_pll_veml_tracking's constructor derives, field by field, and the local replicas equal the block's.
GPU: both chains driven through general_work over the same stream, call for call (window positions, symbol timing, Prompt_I, Doppler, C/N0,
loss of lock, telemetry fault, unusable configurations)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_tracking_adapters")


def _bin():
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        if not g.build_tracking_adapter_test():
            pytest.skip("tests/host/test_tracking_adapters was not prebuilt and /root/reference is not present here")
    return BIN


def test_dll_pll_conf_mapping_equals_reference_constructor():
    r = subprocess.run([_bin(), "conf"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "TRACKING CONF OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.parametrize("launch_ahead,channels_per_handle", [("1", ""), ("0", ""), ("1", "4")])
def test_block_and_runtime_against_the_fake_engine(launch_ahead, channels_per_handle):
    """The whole adapter / block / Hip_Tracking_Runtime stack on the CPU: tests/host/test_tracking_adapters_fake is the same program linked in
    front of tests/host/fake_gsh_engine.cc, a stand-in for the DEVICE half of the C ABI with the oracle's loop behind it (test infrastructure,
    never part of the product).  Every trajectory (13 signals), the loss-of-lock and restart cases, several periods per call, the dump file / TOW /
    time tags and 32 block threads on one shared runtime are compared with the reference's own blocks; the fake engine also checks that every
    push lands at the right absolute sample index and that no two threads are inside one engine handle at once.  Both forms of the runtime's launch
    chain: the next launch queued as soon as one is filed (the default), and one launch, waited for, at a time (GSH_TRK_LAUNCH_AHEAD=0); and once with four
    channels per device handle, so that the 32 blocks of the shared stream sit in eight groups with launches, and wake-ups, of their own."""
    fake = _bin() + "_fake"
    if not os.path.exists(fake):
        pytest.skip("tests/host/test_tracking_adapters_fake was not prebuilt and /root/reference is not present here")
    r = subprocess.run([fake], capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, GSH_TRK_LAUNCH_AHEAD=launch_ahead, **({"GSH_TEST_CHANNELS_PER_LAUNCH": channels_per_handle} if channels_per_handle else {})))
    tail = "\n".join(l for l in r.stdout.splitlines() if "FAIL" in l or "shared stream" in l or "OK" in l or "dump" in l or "restart" in l)
    print(tail[-3000:])
    assert r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout and "FAKE ENGINE" not in r.stderr, tail[-6000:] + r.stderr[-2000:]
    _check_dump_mat("/tmp/gsh_trk_dump_test/hip_trk_ch_6")


def test_launched_mode_against_the_fake_engine():
    """The same program with the runtime's live mode off (GSH_TRK_LIVE=0): one launch per batch of periods, queued ahead, as in round 3 -- the path a
    configuration with <role>.hip_live=false takes."""
    fake = _bin() + "_fake"
    if not os.path.exists(fake):
        pytest.skip("tests/host/test_tracking_adapters_fake was not prebuilt and /root/reference is not present here")
    r = subprocess.run([fake], capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, GSH_TRK_LIVE="0"))
    assert r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout and "FAKE ENGINE" not in r.stderr, r.stdout[-4000:] + r.stderr[-2000:]


def test_blocks_dealt_over_three_devices_share_one_replicated_stream():
    """<role>.hip_devices = 0,1,2 (SURVEY.md 8e behind the adapters): 32 blocks of one role dealt over three devices in turn, ONE Hip_Sample_Ring over the
    engine's stream group -- whichever block is offered new samples first pushes them once, every device's ring receives them --, one runtime (handles,
    residencies) per device.  On the CPU the stand-in engine plays three devices: it checks every ring's pushes against the stream sample for sample, that the
    rings never drift apart, that a loop only ever follows a ring of its own device, and that no two threads are inside one handle; the blocks' windows are
    compared with the reference's own blocks as everywhere else."""
    fake = _bin() + "_fake"
    if not os.path.exists(fake):
        pytest.skip("tests/host/test_tracking_adapters_fake was not prebuilt and /root/reference is not present here")
    r = subprocess.run([fake, "runtime", "32", "2400", "10"], capture_output=True, text=True, timeout=900, cwd="/tmp",
                       env=dict(os.environ, FAKE_GSH_DEVICES="3", GSH_TEST_HIP_DEVICES="0,1,2"))
    print("\n".join(l for l in r.stdout.splitlines() if "shared stream" in l or "FAIL" in l)[-2000:])
    assert r.returncode == 0 and "TRACKING RUNTIME OK" in r.stdout and "FAKE ENGINE" not in r.stderr, r.stdout[-4000:] + r.stderr[-2000:]


def _check_dump_mat(stem):
    """dump_mat (dll_pll_veml_tracking::save_matfile, trk.cc:1706-1890): <dump>.mat holds every field of every 108-byte record of <dump>.dat under the
    reference's variable names and classes (MAT-file level 5, host/hip_mat5_writer.h, read back here with scipy)."""
    import numpy as np
    import scipy.io as sio
    rec = np.dtype([("f0", "<f4", 7), ("PRN_start_sample_count", "<u8"), ("f1", "<f4", 12), ("aux2", "<f8"), ("PRN", "<u4"), ("TOW_ms", "<u8"), ("WN", "<i4")])
    assert rec.itemsize == 108
    d = np.fromfile(stem + ".dat", dtype=rec)
    m = sio.loadmat(stem + ".mat")
    assert len(d) > 100
    names0 = ["abs_VE", "abs_E", "abs_P", "abs_L", "abs_VL", "Prompt_I", "Prompt_Q"]
    names1 = ["acc_carrier_phase_rad", "carrier_doppler_hz", "carrier_doppler_rate_hz", "code_freq_chips", "code_freq_rate_chips", "carr_error_hz",
              "carr_error_filt_hz", "code_error_chips", "code_error_filt_chips", "CN0_SNV_dB_Hz", "carrier_lock_test", "aux1"]
    assert sorted(k for k in m if not k.startswith("__")) == sorted(names0 + names1 + ["PRN_start_sample_count", "aux2", "PRN", "TOW_ms", "WN"])
    for k, nm in enumerate(names0):
        assert m[nm].dtype == np.float32 and m[nm].shape == (1, len(d)) and np.array_equal(m[nm][0].view(np.uint32), d["f0"][:, k].view(np.uint32)), nm
    for k, nm in enumerate(names1):
        assert m[nm].dtype == np.float32 and np.array_equal(m[nm][0].view(np.uint32), np.ascontiguousarray(d["f1"][:, k]).view(np.uint32)), nm
    for nm, dt in (("PRN_start_sample_count", np.uint64), ("aux2", np.float64), ("PRN", np.uint32), ("TOW_ms", np.uint64), ("WN", np.int32)):
        assert m[nm].dtype == dt and np.array_equal(m[nm][0], d[nm]), nm


@pytest.mark.gpu
def test_dropin_throughput_32_channels_one_stream(gpu):
    """BASELINE config 2's shape through the drop-in seam: 32 dll_pll_veml_tracking_hip blocks, one thread each, one 25 Msps stream, one
    Hip_Tracking_Runtime; every block's window positions equal the reference block's.  Prints channel-periods/s (bench.py's `dropin` leg)."""
    import json
    r = subprocess.run([_bin(), "bench", "32", "25000000", "800", "20"], capture_output=True, text=True, timeout=900, cwd="/tmp")
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_JSON")]
    assert r.returncode == 0 and line, r.stdout[-4000:] + r.stderr[-2000:]
    d = json.loads(line[-1][len("DROPIN_JSON"):])
    print(d)
    assert d["windows_within_one_sample"] and d["read_pointers_exact_frac"] >= 0.999 and d["failures"] == 0, d
    assert d["live"] and d["residencies"] >= 1, d   # the default: one resident loop kernel serves all 32 blocks (no launch per batch of periods)
    assert d["channel_periods_per_s"] >= 3.0e5, d  # far above one launch per channel and period; the measured figure sits in bench.py's `dropin` / DESIGN.md


@pytest.mark.gpu
def test_tracking_adapters_follow_the_reference_block(gpu):
    r = subprocess.run([_bin()], capture_output=True, text=True, timeout=900, cwd="/tmp")
    keep = ("FAIL", "periods", "shared stream", "dump", "restart", "noise only", "OK", "failure")
    print("\n".join(l for l in r.stdout.splitlines() if any(k in l for k in keep))[-6000:])
    assert r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout, r.stdout[-6000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_tracking_adapters_in_launched_mode_with_cooperating_work_groups(gpu):
    """<role>.hip_live=false + <role>.hip_work_groups_per_channel=2 (here through the runtime's test switches): every handle the runtime opens runs its launches
    with two work-groups per channel (gsh_trk_set_split); the thirteen signals' trajectories, restarts, dump files and the 32 blocks of the shared stream are
    compared with the reference's own blocks as in the default mode -- the sums differ from the one-work-group form's by rounding only."""
    r = subprocess.run([_bin()], capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, GSH_TRK_LIVE="0", GSH_TRK_WORK_GROUPS="2"))
    keep = ("FAIL", "periods", "shared stream", "OK", "failure")
    print("\n".join(l for l in r.stdout.splitlines() if any(k in l for k in keep))[-4000:])
    assert r.returncode == 0 and "TRACKING ADAPTERS OK" in r.stdout, r.stdout[-6000:] + r.stderr[-2000:]


# ---- the reference's own Channel / ChannelFsm / channel_msg_receiver_cc over the HIP adapters (tests/host/test_channel.cc, round 5) ---------------------------------------------
CHAN = os.path.join(ROOT, "tests", "host", "test_channel")


def _chan(fake):
    exe = CHAN + ("_fake" if fake else "")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        if not g.build_channel_test():
            pytest.skip("tests/host/test_channel was not prebuilt and /root/reference is not present here")
    return exe


def _run_chan(exe, *args, timeout=900):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, cwd="/tmp")
    keep = ("FAIL", "channel life", "hand-over", "churn:", "fault ", "CHANNEL", "failure")
    print("\\n".join(l for l in r.stdout.splitlines() if any(k in l for k in keep))[-6000:])
    assert r.returncode == 0 and "CHANNEL OK" in r.stdout and "FAKE ENGINE" not in r.stderr, r.stdout[-5000:] + r.stderr[-2000:]
    return r.stdout


def test_channel_life_reference_channel_over_hip_adapters_fake_engine():
    """north_star: "drops into a Channel unchanged".  The reference's Channel, ChannelFsm and channel_msg_receiver_cc (compiled in place) over GpsL1CaPcpsAcquisitionHip +
    GpsL1CaDllPllTrackingHip, one thread per block, token-scheduled: acquisition of an absent satellite fails ("events" 2 -> FSM -> queue -> control assigns the next one),
    the present one is acquired, the acquisition block's thread calls ChannelFsm::Event_valid_acquisition directly -> start_tracking, the signal is removed -> lock detectors
    -> "events" 3 -> channel_msg_receiver_cc -> FSM standby -> queue -> re-acquisition attempts until the signal is back -> tracked to the end.  The same Channel class over
    the reference's own adapters runs the same stream: every event (who, what, satellite, source position), every hand-over (Acq_delay_samples, Acq_doppler_hz,
    Acq_samplestamp_samples), the acquisition block's consumed counts call for call, the tracking block's read pointers and every item the telemetry decoder receives are
    compared.  CPU: the fake engine (oracle loop, a double-precision PCPS search) stands in for the device."""
    out = _run_chan(_chan(True), "life")
    assert "events identical" in out


def test_channel_churn_fake_engine():
    """Twelve channels on one stream, free-running block threads, four of them forced to lose lock every 200 ms of stream (telemetry fault message) and re-acquired through
    the FSM while eight track: start_tracking / stop_tracking of a churner quiesce the live residency they all share.  The steady channels keep every window, publish no
    event and stop at the reference receiver's read pointers."""
    _run_chan(_chan(True), "churn")


def test_engine_failures_surface_as_channel_events_fake_engine():
    """an engine failure must reach the Channel as loss of lock ("events" 3) or a failed acquisition ("events" 2), never as an exception across general_work
    or a hung thread.  Injected into the fake engine: a push fails once / for good, gsh_trk_live_take fails for one channel, a residency never reports (the runtime's record
    watchdog), a dwell fails, gsh_trk_start fails, a launch fails in launched mode."""
    _run_chan(_chan(True), "faults")


def test_acquisition_and_tracking_blocks_of_a_channel_are_dealt_to_the_same_gpu():
    """<role>.hip_devices = 0,1,2 on both roles: channel c's acquisition block and tracking block land on GPU c mod 3; hip_device pins.  The fake engine plays three devices."""
    exe = _chan(True)
    r = subprocess.run([exe, "devices"], capture_output=True, text=True, timeout=300, cwd="/tmp", env=dict(os.environ, FAKE_GSH_DEVICES="3"))
    assert r.returncode == 0 and "dealt 0 1 2 0 1 2 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_channel_life_and_churn_on_the_gpu(gpu):
    """The same program against libgnss_sdr_hip.so: the channel life token-scheduled beside the reference receiver (identical events, hand-overs, consumed counts), then BASELINE
    config 2's 32 channels on one stream with 8 of them churning through the FSM while 24 track on the shared live residency (prints the stall per start / stop)."""
    exe = _chan(False)
    _run_chan(exe, "life")
    _run_chan(exe, "churn", "32", "8", "2.4", "1", timeout=1500)
