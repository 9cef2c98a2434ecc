"""GPU tests of the live mode of the device-resident DLL/PLL loop (gsh_trk_live_*): the kernel stays resident, follows the ring as pushes complete and
leaves its records in page-locked host memory -- no launch per batch of periods (what dll_pll_veml_tracking's one-period-per-general_work cadence,
trk.cc:1898-2001, would otherwise cost).  Same kernel, same arithmetic: the records must equal those of gsh_trk_run over the flat stream BYTE FOR BYTE
(which tests/test_tracking_loop_gpu.py in turn holds against the oracle loop)."""
import time

import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream

pytestmark = pytest.mark.gpu

FS, N = 2.046e6, 2046


def _loop(gpu, conf_kw, n_channels, max_len=1023):
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    return TrackingLoop(trk_conf(**conf_kw), n_channels, max_len, device=gpu)


def _scenario(epochs, seed=77):
    prns, dops, cphs = [5, 18], [-1900.0, 3300.0], [100.0, 900.5]
    total = (epochs + 3) * N
    x = synth_gps_l1_stream(total, FS, prns, dops, cphs, cn0_dbhz=47.0, seed_noise=seed)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 25.0), -127, 127).astype(np.int8)
    xf = (x8[:, 0].astype(np.float32) + 1j * x8[:, 1].astype(np.float32)).astype(np.complex64)
    starts = []
    for fd, cph in zip(dops, cphs):
        f_code = 1.023e6 * (1 + fd / 1575.42e6)
        starts.append(int(round((1023.0 - cph) / f_code * FS)))
    # a third channel searches a satellite that is not there: with the lock detectors on it is declared lost after a while
    prns, dops, starts = prns + [27], dops + [500.0], starts + [1234]
    return prns, dops, starts, total, x8, xf


# (pull_in_time_s = 0 still means one whole second of pull-in -- the reference's integer arithmetic, trk.cc:1912 -- so a loss of lock needs > 1000 periods)
KW = dict(fs_in=FS, vector_length=N, pll_bw_hz=25.0, dll_bw_hz=2.0, enable_lock_detectors=1, pull_in_time_s=0, cn0_min=32, max_code_lock_fail=50)


def _flat_run(gpu, prns, dops, starts, xf, epochs):
    flat = _loop(gpu, KW, n_channels=len(prns))
    flat.set_stream_host(xf)
    for ch in range(len(prns)):
        flat.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    rec, done = flat.run(epochs + 10)
    flat.close()
    return rec, done


def _bytes(records):
    return b"".join(bytes(memoryview(r)) for r in records)


def _drain(live, got, lost, deadline_s, want=None):
    """poll the channels' record rings until nothing more arrives for a while (or `want` records per running channel are there)"""
    t_end = time.time() + deadline_s
    quiet_since = time.time()
    while time.time() < t_end:
        progress = False
        for ch in range(len(got)):
            if lost[ch]:
                continue
            rec, pending, nw, active = live.live_take(ch, 64)
            if rec:
                got[ch] += rec
                progress = True
                if rec[-1].flags & 2:
                    lost[ch] = True
        if progress:
            quiet_since = time.time()
        elif want is not None and all(lost[ch] or len(got[ch]) >= want[ch] for ch in range(len(got))):
            return
        elif time.time() - quiet_since > 0.05 and live.live_in_flight() == 0:
            return
        else:
            time.sleep(0.0005)


@pytest.mark.parametrize("item", ["ibyte", "gr_complex"])
def test_live_records_equal_launched_runs(gpu, item):
    from gnss_sdr_amd.sample_stream import SampleStream
    epochs = 1150
    prns, dops, starts, total, x8, xf = _scenario(epochs)
    rec_flat, done_flat = _flat_run(gpu, prns, dops, starts, xf, epochs)
    assert done_flat[0] >= epochs - 2 and 1000 < done_flat[2] < epochs - 20, done_flat  # two channels track to the end, the third loses lock on the way
    assert rec_flat[2][done_flat[2] - 1].flags & 2

    ring = SampleStream(23 * N + 7, 2 * N, device=gpu)  # much shorter than the stream: it wraps a dozen times
    live = _loop(gpu, KW, n_channels=3)
    live.set_stream_ring(ring)
    for ch in range(3):
        live.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    got, lost = [[], [], []], [False, False, False]
    pushed, blk, begins = 0, 9 * N + N // 2, 0
    src = x8 if item == "ibyte" else xf
    while pushed < total:
        m = min(blk, total - pushed)
        ring.push(src[pushed:pushed + m], item)
        pushed += m
        if live.live_in_flight() == 0:
            live.live_begin()
            begins += 1
        # what this block lets every running channel do: wait for it (the device needs microseconds; Python is the slow side)
        want = [min(done_flat[ch], max(0, (pushed - starts[ch]) // N - 1)) for ch in range(3)]
        _drain(live, got, lost, 2.0, want)
    _drain(live, got, lost, 1.0)
    live.live_quiesce()
    _drain(live, got, lost, 0.2)
    for ch in range(3):
        assert len(got[ch]) == done_flat[ch], (ch, len(got[ch]), done_flat[ch])
        assert _bytes(got[ch]) == _bytes(rec_flat[ch][:done_flat[ch]]), f"channel {ch}: live records differ from the launched run"
    assert lost == [False, False, True]
    assert abs(np.mean([r.carrier_doppler_hz for r in got[0][-60:]]) - dops[0]) < 2.0
    # far fewer residencies than blocks would be possible, but Python dawdles between pushes: each residency idles out (200 us); what matters
    # is that every one of them picked the channels up exactly where the previous one left them
    assert 1 <= begins <= total // blk + 2
    live.close()
    ring.close()


def test_one_residency_follows_many_pushes(gpu):
    """A residency that is given time (idle timeout 200 ms) serves push after push without another launch; the second one queued behind it takes over
    when the first has used up its residency time."""
    from gnss_sdr_amd.sample_stream import SampleStream
    epochs = 120
    prns, dops, starts, total, x8, xf = _scenario(epochs, seed=5)
    rec_flat, done_flat = _flat_run(gpu, prns, dops, starts, xf, epochs)
    ring = SampleStream(40 * N, 2 * N, device=gpu)
    live = _loop(gpu, KW, n_channels=3)
    live.set_stream_ring(ring)
    live.live_configure(idle_timeout_us=200000, residency_us=300000)
    for ch in range(3):
        live.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    live.live_begin()
    live.live_begin()  # the second waits behind the first
    assert live.live_in_flight() == 2
    live.live_begin()  # a third is not queued
    assert live.live_in_flight() == 2
    got, lost = [[], [], []], [False, False, False]
    pushed, blk = 0, 3 * N + 17
    t0 = time.time()
    while pushed < total:
        m = min(blk, total - pushed)
        ring.push(xf[pushed:pushed + m])
        pushed += m
        want = [min(done_flat[ch], max(0, (pushed - starts[ch]) // N - 1)) for ch in range(3)]
        _drain(live, got, lost, 2.0, want)
    assert live.live_in_flight() >= 1 or time.time() - t0 > 0.25  # still the residencies queued at the start (unless the box was very slow)
    live.live_quiesce()
    assert live.live_in_flight() == 0
    _drain(live, got, lost, 0.2)
    for ch in range(3):
        assert _bytes(got[ch]) == _bytes(rec_flat[ch][:done_flat[ch]]), f"channel {ch}"
    live.close()
    ring.close()


def test_live_rules(gpu):
    """start / stop / run need the device quiet; a push may not overwrite what a live channel still has to read; an idle residency leaves by itself;
    a stopped and restarted channel carries on with fresh records; launches and residencies may alternate on one handle."""
    from gnss_sdr_amd import _lib
    from gnss_sdr_amd.sample_stream import SampleStream
    epochs = 60
    prns, dops, starts, total, x8, xf = _scenario(epochs, seed=9)
    ring = SampleStream(12 * N, 2 * N, device=gpu)
    live = _loop(gpu, dict(KW, enable_lock_detectors=0), n_channels=2)
    live.set_stream_ring(ring)
    live.start(0, oracle.ca_code(prns[0]), starts[0], 0, dops[0] + 6.0)
    live.live_configure(idle_timeout_us=100000, residency_us=1000000)
    ring.push(xf[:8 * N])
    live.live_begin()
    with pytest.raises(_lib.GshError):
        live.start(1, oracle.ca_code(prns[1]), starts[1], 0, dops[1])  # a residency is in flight
    with pytest.raises(_lib.GshError):
        live.run(1)
    got, lost = [[], []], [False, True]
    _drain(live, got, lost, 2.0, want=[6, 0])
    assert len(got[0]) >= 6
    # the channel stands at ~7 periods; the ring holds 12: a push that would wrap over its next window is refused, one that just fits is not
    nw = live.live_take(0, 0)[2]
    assert nw >= starts[0] + 6 * N
    room = nw + 12 * N - 8 * N
    with pytest.raises(_lib.GshError):
        ring.push(xf[8 * N:8 * N + room + 2 * N])
    ring.push(xf[8 * N:8 * N + room - 8])
    pushed = 8 * N + room - 8
    assert live.live_in_flight() == 1, "the refused push must not have waited for the device (an outgrown staging buffer is parked, not freed)"
    _drain(live, got, lost, 2.0, want=[(pushed - starts[0]) // N - 1, 0])
    live.live_quiesce()
    _drain(live, got, lost, 0.2)
    n_before = len(got[0])
    # launches and residencies alternate: two periods by gsh_trk_run, then live again
    ring.push(xf[pushed:pushed + 3 * N])
    pushed += 3 * N
    rec, done = live.run(2)
    assert done[0] == 2 and rec[0][0].sample_counter == got[0][-1].sample_counter + got[0][-1].prn_length_samples
    # stop, start another satellite on the other channel in the quiet
    live.start(1, oracle.ca_code(prns[1]), rec[0][1].sample_counter + N, 0, dops[1])
    live.live_configure(idle_timeout_us=300, residency_us=5000)
    live.live_begin()
    t0 = time.time()
    while live.live_in_flight() and time.time() - t0 < 2.0:
        time.sleep(0.001)
    assert live.live_in_flight() == 0, "an idle residency did not leave by itself"
    lost = [False, False]
    _drain(live, got, lost, 0.2)
    assert len(got[0]) == n_before + 0 or got[0][n_before].sample_counter == rec[0][1].sample_counter + rec[0][1].prn_length_samples
    live.close()
    ring.close()
