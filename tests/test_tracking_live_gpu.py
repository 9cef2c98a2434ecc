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


# ---- the live flavours at the BASELINE shapes, DIRECTLY: the records of a residency are held to the launched run byte for byte AND to the oracle loop in the same
# test (round 4 reached the LIVE kernels at N = 25 000 and N = 128 000 / five taps + pilot only through the adapter program, and the oracle only transitively)
def _shape(name):
    from helpers import add_code_signal, cn0_to_amplitude, golden_e1_l5_codes
    if name == "config2_25Msps_3taps":
        fs, n, epochs = 25e6, 25000, 210
        prns, dops, cphs = [3, 22], [2750.0, -4100.0], [17.25, 640.0]
        x = synth_gps_l1_stream((epochs + 3) * n, fs, prns, dops, cphs, cn0_dbhz=46.0, seed_noise=91)
        kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, enable_lock_detectors=1, early_late_space_chips=0.5)
        codes = [(oracle.ca_code(p), None) for p in prns]
        starts = [int(round((1023.0 - cph) / (1.023e6 * (1 + fd / 1575.42e6)) * fs)) for fd, cph in zip(dops, cphs)]
        hand_over = [fd + 7.0 for fd in dops]
        return dict(fs=fs, n=n, epochs=epochs, x=x, kw=kw, codes=codes, starts=starts, dops=hand_over, taps=3, max_len=1023, scale=20.0)
    # BASELINE config 4: Galileo E1, 32 Msps, 4 ms windows of 128 000 samples, VE/E/P/L/VL on the pilot + the data prompt (trk.cc:1246-1256)
    fs, n, epochs = 32e6, 128000, 203
    g = golden_e1_l5_codes()
    rng = np.random.default_rng(43)
    n_stream = (epochs + 3) * n
    x = (rng.standard_normal(n_stream, dtype=np.float32) + 1j * rng.standard_normal(n_stream, dtype=np.float32)).astype(np.complex64)
    amp = cn0_to_amplitude(45.0, fs)
    sig = [(7, -1830.0, 3000.0), (19, 2410.0, 511.0)]
    starts, hand_over, codes = [], [], []
    for prn_i, fd, ph in sig:
        rate = 1.023e6 * (1 + fd / 1575.42e6) / fs * 2.0
        add_code_signal(x, (g["e1b"][prn_i] - g["e1c"][prn_i]) / np.sqrt(2.0), fs, rate, ph, fd, amp)
        starts.append(int(round((8184.0 - ph) / rate)))
        hand_over.append(fd - 5.0)
        codes.append((g["e1c"][prn_i], g["e1b"][prn_i]))
    kw = dict(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, cloop=0, early_late_space_chips=0.15,
              very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pll_filter_order=3, dll_filter_order=2, enable_lock_detectors=1)
    return dict(fs=fs, n=n, epochs=epochs, x=x, kw=kw, codes=codes, starts=starts, dops=hand_over, taps=5, max_len=8184, scale=20.0)


@pytest.mark.parametrize("shape", ["config2_25Msps_3taps", "config4_e1_32Msps_5taps_pilot"])
def test_live_at_the_baseline_shapes_equals_launched_run_and_oracle(gpu, shape):
    from gnss_sdr_amd.sample_stream import SampleStream
    from test_tracking_loop_gpu import _compare
    s = _shape(shape)
    n, epochs = s["n"], s["epochs"]
    # the front-end's 8-bit items, as the ring receives them; the flat run and the oracle read the same values as complex64
    x8 = np.clip(np.round(np.stack([s["x"].real, s["x"].imag], axis=1) * s["scale"]), -127, 127).astype(np.int8)
    xf = (x8[:, 0].astype(np.float32) + 1j * x8[:, 1].astype(np.float32)).astype(np.complex64)
    total = len(xf)
    nch = len(s["codes"])
    flat = _loop(gpu, s["kw"], n_channels=nch, max_len=s["max_len"])
    flat.set_stream_host(xf)
    for ch in range(nch):
        flat.start(ch, s["codes"][ch][0], s["starts"][ch], 0, s["dops"][ch], data_code=s["codes"][ch][1])
    rec_flat, done_flat = flat.run(epochs)
    flat.close()
    assert all(d >= 200 for d in done_flat), done_flat

    ring = SampleStream(9 * n + 7, 2 * n, device=gpu)   # shorter than the stream: it wraps
    live = _loop(gpu, s["kw"], n_channels=nch, max_len=s["max_len"])
    live.set_stream_ring(ring)
    for ch in range(nch):
        live.start(ch, s["codes"][ch][0], s["starts"][ch], 0, s["dops"][ch], data_code=s["codes"][ch][1])
    got, lost = [[] for _ in range(nch)], [False] * nch
    pushed, blk = 0, 3 * n + n // 2
    while pushed < total:
        m = min(blk, total - pushed)
        ring.push(x8[pushed:pushed + m], "ibyte")
        pushed += m
        if live.live_in_flight() == 0:
            live.live_begin()
        want = [min(done_flat[ch], max(0, (pushed - s["starts"][ch]) // n - 1)) for ch in range(nch)]
        _drain(live, got, lost, 3.0, want)
    _drain(live, got, lost, 1.0)
    live.live_quiesce()
    _drain(live, got, lost, 0.2)
    live.close()
    ring.close()
    conf_o = oracle.trk_conf(**s["kw"])
    for ch in range(nch):
        # (the launched run was asked for `epochs` periods; the residency goes on to the last window the stream holds)
        assert done_flat[ch] <= len(got[ch]) <= done_flat[ch] + 4, (shape, ch, len(got[ch]), done_flat[ch])
        assert _bytes(got[ch][:done_flat[ch]]) == _bytes(rec_flat[ch][:done_flat[ch]]), f"{shape} channel {ch}: live records differ from the launched run"
        # ... and the first 200 periods of the LIVE records against the oracle loop, the same bars as the launched loop's (test_tracking_loop_gpu._compare)
        ora = oracle.trk_run(conf_o, s["codes"][ch][0], xf, s["starts"][ch], 0, s["dops"][ch], 200, data_code=s["codes"][ch][1])
        assert len(ora) == 200
        stats = _compare(got[ch][:200], ora, s["taps"], f"{shape} live ch{ch}", xmax=6.0 * s["scale"])
        assert stats["compared"] == 200
        tail = got[ch][150:200]
        assert abs(np.mean([r.carrier_doppler_hz for r in tail]) - (s["dops"][ch] + (-7.0 if s["taps"] == 3 else 5.0))) < 3.0, (shape, ch)


def test_ring_destroyed_under_a_live_residency(gpu):
    """SURVEY section 5 / round-4 verdict: an engine failure must not crash the receiver.  gsh_stream_destroy of the ring a residency is following: the ring tells the loop's
    handle first (the resident kernel leaves, the registration goes), THEN frees its memory.  The handle is left without a stream -- gsh_trk_live_begin fails with
    GSH_ERR_STATE, which a tracking block turns into "events" 3 --; the records the channels had finished can still be taken; and with a new ring, positioned where the
    slowest channel stands, the very same handle goes on: all records byte-identical to one launched run over the flat stream.  Also: gsh_stream_seek is refused while a
    live channel still reads the ring (a seek under a resident kernel could let a push overwrite the window it is correlating)."""
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.sample_stream import SampleStream
    epochs = 500
    prns, dops, starts, total, x8, xf = _scenario(epochs)
    prns, dops, starts = prns[:2], dops[:2], starts[:2]      # the two satellites that are there
    rec_flat, done_flat = _flat_run(gpu, prns, dops, starts, xf, epochs)
    ring = SampleStream(200 * N + 7, 2 * N, device=gpu)
    live = _loop(gpu, KW, n_channels=2)
    live.set_stream_ring(ring)
    for ch in range(2):
        live.start(ch, oracle.ca_code(prns[ch]), starts[ch], 0, dops[ch] + 6.0)
    live.live_configure(idle_timeout_us=500000, residency_us=5000000)   # the residency stays for half a second after its last period: it IS resident when the ring goes
    got, lost = [[], []], [False, False]
    first = 180 * N
    ring.push(x8[:first], "ibyte")
    live.live_begin()
    _drain(live, got, lost, 3.0, [150, 150])
    assert min(len(g) for g in got) >= 150 and live.live_in_flight() >= 1
    with pytest.raises(GshError):
        ring.seek(0)                      # refused: live channels still read the ring
    ring.close()                          # <- the ring goes under the resident kernel
    assert live.live_in_flight() == 0
    with pytest.raises(GshError):
        live.live_begin()                 # no stream any more: GSH_ERR_STATE
    _drain(live, got, lost, 0.3)          # what had been finished is still there
    nxt = [live.live_take(ch, 1)[2] for ch in range(2)]
    assert all(len(got[ch]) >= 150 for ch in range(2)) and lost == [False, False]
    # a new ring where the slowest channel stands; the handle goes on from its own state
    ring2 = SampleStream(200 * N + 7, 2 * N, device=gpu)
    at = min(nxt)
    ring2.seek(at)
    live.set_stream_ring(ring2)
    pushed = at
    while pushed < total:
        m = min(9 * N, total - pushed)
        ring2.push(x8[pushed:pushed + m], "ibyte")
        pushed += m
        if live.live_in_flight() == 0:
            live.live_begin()
        _drain(live, got, lost, 2.0, [min(done_flat[ch], max(0, (pushed - starts[ch]) // N - 1)) for ch in range(2)])
    _drain(live, got, lost, 0.5)
    live.live_quiesce()
    _drain(live, got, lost, 0.2)
    for ch in range(2):
        assert done_flat[ch] <= len(got[ch]) <= done_flat[ch] + 4, (ch, len(got[ch]), done_flat[ch])
        assert _bytes(got[ch][:done_flat[ch]]) == _bytes(rec_flat[ch][:done_flat[ch]]), f"channel {ch}: records differ from the launched run after the ring was replaced"
    live.close()
    ring2.close()
