"""The N > 1 path of the block-replication group (gsh_stream_group_*, csrc/stream_group.hip; SURVEY.md 8e: "RCCL broadcast of the shared input sample
block over xGMI"), executed in two transports:

  * "rccl"  -- the system's RCCL across >= 2 visible GPUs (the driver's 8-GPU node); skipped on a one-GPU box, where RCCL refuses two ranks on one device;
  * "stub"  -- tests/host/libfake_rccl.so (tests/host/fake_rccl.cc, TEST INFRASTRUCTURE, selected through GSH_RCCL_LIBRARY) standing in for librccl.so: the
               engine's own code -- gsh_stream_group_plan's chunk offsets and padded tails, the root-local index, the scatter's send-to-self, the two staging
               slots under back-to-back pushes, the event ordering against the cast and the readers -- runs with 3 ranks (not a power of two) on ONE device.
               The stand-in is stream-ordered and asynchronous within a process, file-backed between processes; it proves the plan, not the wire.

  * one process driving all ranks (gsh_stream_group_create, ncclCommInitAll): after every push, every rank's ring holds bit for bit what a
    local gsh_stream_push of the same 8-bit items leaves on that device, in both group modes (broadcast / scatter + all-gather);
  * a correlator bank bound to each rank's ring returns exactly what a bank on a private ring of that device returns;
  * one process per rank (gsh_stream_group_create_rank, the layout bench.py --gpus N uses): ranks are fresh interpreters (multiprocessing, spawn), the
    communicator id travels through a file, rank 0 supplies the blocks, every rank checks its own ring against the items;
  * closed loops and acquisition PRN shards on every ring of the group;
  * bench.py --gpus 2 end to end (ranks spawned by bench.py itself)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream, tracking_params_for

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_RCCL = os.path.join(ROOT, "tests", "host", "libfake_rccl.so")
STUB_RANKS = 3


def _n_devices() -> int:
    import gnss_sdr_amd
    return int(gnss_sdr_amd.load().gsh_device_count())


@pytest.fixture(params=["rccl", "stub"])
def devices(request, monkeypatch, gpu):
    """The device of every rank of the group under test: rank i -> devices[i]."""
    n = _n_devices()
    if request.param == "rccl":
        if n < 2:
            pytest.skip(f"{n} HIP device(s) visible: real RCCL needs one device per rank (the stub transport covers this box)")
        monkeypatch.delenv("GSH_RCCL_LIBRARY", raising=False)
        return list(range(min(n, 8)))
    assert os.path.exists(FAKE_RCCL), "tests/host/libfake_rccl.so was not built (__graft_entry__.build)"
    monkeypatch.setenv("GSH_RCCL_LIBRARY", FAKE_RCCL)
    from gnss_sdr_amd.sample_stream import StreamGroup
    assert os.path.samefile(StreamGroup.library(), FAKE_RCCL)
    return [r % n for r in range(STUB_RANKS)]


def _blocks(rng, sizes):
    return [rng.integers(-128, 128, size=(n, 2)).astype(np.int8) for n in sizes]


@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_every_ring_of_the_group_equals_a_local_push(devices, mode):
    n_dev = len(devices)
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    cap, win = 40000, 9000
    g = StreamGroup.local(devices, cap, win, mode=mode)
    assert g.size() == n_dev
    refs = [SampleStream(cap, win, device=d) for d in devices]
    rng = np.random.default_rng(5)
    total = 0
    for a in _blocks(rng, (9000, 1, 8191, 20000, 0, 777, 33333, 4096, n_dev * 1000 + 3)):  # ragged, empty, not a multiple of the group size, wraps
        first = g.push(a, len(a), "ibyte")
        assert first == total
        for r in refs:
            assert r.push(a, "ibyte") == total
        total += len(a)
        g.wait()
        for i, d in enumerate(devices):
            ring = g.ring(i)
            lo, hi = ring.range()
            assert (lo, hi) == refs[i].range() == (max(0, total - cap), total)
            n = min(hi - lo, win)
            for start in (lo, hi - n):
                assert np.array_equal(ring.read(start, n).view(np.uint32), refs[i].read(start, n).view(np.uint32)), (mode, d, start)
    g.close()


def test_banks_on_every_device_of_the_group_match_private_rings(devices):
    n_dev = len(devices)
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    from gnss_sdr_amd.tracking import CorrelatorBank
    fs, n = 4e6, 4000
    total = 24 * n
    dopplers = [1000.0, -2000.0, 300.0]
    x = synth_gps_l1_stream(total, fs, [1, 2, 3], dopplers, [5.0, 300.0, 800.0], seed_noise=21)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 30.0), -127, 127).astype(np.int8)
    cap = 9 * n + 2
    g = StreamGroup.local(devices, cap, 2 * n, mode="scatter_allgather")
    rng = np.random.default_rng(2)
    params = [tracking_params_for(fs, d, rng) for d in dopplers]
    banks = []
    for i, d in enumerate(devices):
        b = CorrelatorBank(3, 1023, device=d)
        for c in range(3):
            b.set_code(c, oracle.ca_code(c + 1))
        b.set_stream_ring(g.ring(i))
        banks.append(b)
    group_out = [[] for _ in devices]
    job_lists = []
    for blk in range(0, total, 3 * n + 17):
        m = min(3 * n + 17, total - blk)
        g.push(x8[blk:blk + m], m, "ibyte")  # no host wait between the collective and the launches that read its result
        lo, hi = g.ring(0).range()
        jobs = [dict(sample_offset=hi - n - k * (n // 2 + 3) - c, n_samples=n, code_slot=c, shifts_chips=[-0.5, 0.0, 0.5], **params[c])
                for c in range(3) for k in range(3) if hi - n - k * (n // 2 + 3) - c >= lo]
        job_lists.append((blk, m, jobs))
        for i, b in enumerate(banks):
            if jobs:
                group_out[i].append(b.correlate(jobs))
    for b in banks:
        b.close()
    g.close()
    # the same job lists on a private ring per device
    for i, d in enumerate(devices):
        priv = SampleStream(cap, 2 * n, device=d)
        b = CorrelatorBank(3, 1023, device=d)
        for c in range(3):
            b.set_code(c, oracle.ca_code(c + 1))
        b.set_stream_ring(priv)
        got = []
        for blk, m, jobs in job_lists:
            priv.push(x8[blk:blk + m], "ibyte")
            if jobs:
                got.append(b.correlate(jobs))
        b.close()
        a, e = np.concatenate(group_out[i], axis=0), np.concatenate(got, axis=0)
        assert a.shape == e.shape and np.array_equal(a.view(np.uint32), e.view(np.uint32)), f"device {d}"


def _rank_main(rank, world, device, id_path, mode, result_path):
    """One process per GPU, as bench.py --gpus N runs: gsh_stream_group_create_rank with a communicator id from rank 0."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import time
    from gnss_sdr_amd.sample_stream import StreamGroup
    if rank == 0:
        uid = StreamGroup.unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
    else:
        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 120:
                raise RuntimeError("no communicator id from rank 0")
            time.sleep(0.01)
        uid = open(id_path, "rb").read()
    cap, win = 40000, 9000
    g = StreamGroup.from_rank(device, rank, world, uid, cap, win, mode=mode)
    ring = g.ring(0)
    rng = np.random.default_rng(11)  # every rank draws the same blocks: rank 0 pushes them, the others use them to check their ring
    ok, total = True, 0
    for a in _blocks(rng, (9000, 8191, 20000, 777, 33333, world * 1000 + 3)):
        first = g.push(a if rank == 0 else None, len(a), "ibyte")
        ok = ok and first == total
        total += len(a)
        g.wait()
        lo, hi = ring.range()
        ok = ok and (lo, hi) == (max(0, total - cap), total)
        m = min(len(a), win)
        want = (a[-m:, 0].astype(np.float32) + 1j * a[-m:, 1].astype(np.float32)).astype(np.complex64)
        ok = ok and bool(np.array_equal(ring.read(hi - m, m), want))
    g.close()
    with open(f"{result_path}.{rank}", "w") as f:
        f.write("ok" if ok else "MISMATCH")


@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_one_process_per_gpu_group(devices, mode):
    n_dev = len(devices)
    import multiprocessing as mp
    world = min(n_dev, 8)
    ctx = mp.get_context("spawn")  # fresh interpreters: one HIP runtime per rank, nothing inherited from the pytest process
    with tempfile.TemporaryDirectory() as tmp:
        id_path, result_path = os.path.join(tmp, "nccl_id"), os.path.join(tmp, "result")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        procs = [ctx.Process(target=_rank_main, args=(r, world, devices[r], id_path, mode, result_path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
        for r, p in enumerate(procs):
            if p.is_alive():
                p.kill()
                pytest.fail(f"rank {r} ({mode}) did not finish")
            assert p.exitcode == 0, f"rank {r} ({mode}) exited with {p.exitcode}"
            assert open(f"{result_path}.{r}").read() == "ok", f"rank {r} ({mode})"


def _trk_scenario(fs, n, epochs):
    prns, dops, cphs = [5, 18, 9, 27], [-1900.0, 3300.0, 700.0, -3100.0], [100.0, 900.5, 333.0, 12.0]
    total = (epochs + 3) * n
    x = synth_gps_l1_stream(total, fs, prns, dops, cphs, cn0_dbhz=47.0, seed_noise=31)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 25.0), -127, 127).astype(np.int8)
    starts = [int(round((1023.0 - cph) / (1.023e6 * (1 + fd / 1575.42e6)) * fs)) for fd, cph in zip(dops, cphs)]
    return prns, dops, starts, total, x8


def _records_bytes(recs):
    return b"".join(bytes(memoryview(r)) for r in recs)


@pytest.mark.parametrize("live", [False, True])
def test_closed_loops_on_every_ring_of_the_group_match_private_rings(devices, live):
    """SURVEY 8e for the path a receiver runs: the DLL/PLL loop closed on the device (gsh_trk_*), channel c on GPU c mod G, every GPU's loop bound to ITS ring of
    the group -- launched after every replicated block, and as live residencies that follow the ring.  The records of every channel must equal, byte for byte, those
    of the same channel on a private ring of the same GPU fed with the same items."""
    n_dev = len(devices)
    import time
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    fs, n, epochs = 4e6, 4000, 120
    prns, dops, starts, total, x8 = _trk_scenario(fs, n, epochs)
    kw = dict(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=3.0, enable_lock_detectors=1, pull_in_time_s=0)
    cap, win = 40 * n, 2 * n
    blk = 7 * n + 11

    def run(rings, push, wait):
        loops, chan_of = [], []
        for i, d in enumerate(devices):
            mine = [c for c in range(len(prns)) if c % n_dev == i]
            lp = TrackingLoop(trk_conf(**kw), max(len(mine), 1), 1023, device=d)
            lp.set_stream_ring(rings[i])
            for k, c in enumerate(mine):
                lp.start(k, oracle.ca_code(prns[c]), starts[c], 0, dops[c] + 6.0)
            if live:
                lp.live_configure(idle_timeout_us=200000, residency_us=2000000)
            loops.append(lp)
            chan_of.append(mine)
        got = {c: [] for c in range(len(prns))}
        pushed = 0
        if live:
            for lp in loops:
                lp.live_begin()
        while pushed < total:
            m = min(blk, total - pushed)
            push(x8[pushed:pushed + m], m)
            pushed += m
            if not live:
                for i, lp in enumerate(loops):
                    rec, done = lp.run(10)  # (no host wait between the collective and the launch that reads its result: the ring's events order them)
                    for k, c in enumerate(chan_of[i]):
                        got[c] += rec[k]
            else:
                want = {c: max(0, (pushed - starts[c]) // n - 1) for c in got}
                t_end = time.time() + 5.0
                while time.time() < t_end and any(len(got[c]) < want[c] for c in got):
                    for i, lp in enumerate(loops):
                        for k, c in enumerate(chan_of[i]):
                            got[c] += lp.live_take(k, 64)[0]
        if live:
            for i, lp in enumerate(loops):
                lp.live_quiesce()
                for k, c in enumerate(chan_of[i]):
                    got[c] += lp.live_take(k, 256)[0]
        wait()
        for lp in loops:
            lp.close()
        return got

    g = StreamGroup.local(devices, cap, win, mode="scatter_allgather")
    group_got = run([g.ring(i) for i in range(n_dev)], lambda a, m: g.push(a, m, "ibyte"), g.wait)
    g.close()
    priv = [SampleStream(cap, win, device=d) for d in devices]

    def push_all(a, m):
        for r in priv:
            r.push(a, "ibyte")

    priv_got = run(priv, push_all, lambda: None)
    for c in range(len(prns)):
        assert len(group_got[c]) >= epochs - 12, (c, len(group_got[c]))
        assert _records_bytes(group_got[c]) == _records_bytes(priv_got[c]), f"channel {c} (device {c % n_dev}, live={live})"
        assert abs(np.mean([r.carrier_doppler_hz for r in group_got[c][-40:]]) - dops[c]) < 3.0


def test_acquisition_prn_shards_on_every_ring_of_the_group(devices):
    """"acquisition: PRN p -> GPU p mod G" -- every GPU searches its share of the PRNs over the SAME replicated block (gsh_acq_dwell_ring on its
    ring of the group); results equal those of the same search over a private ring of that GPU, and the union finds every embedded satellite."""
    n_dev = len(devices)
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    fs, n = 4e6, 4000
    sig_prns, dops, cphs = [3, 8, 14, 22, 30], [1500.0, -3250.0, 250.0, 4000.0, -750.0], [10.0, 444.0, 901.5, 77.0, 600.0]
    x = synth_gps_l1_stream(6 * n, fs, sig_prns, dops, cphs, cn0_dbhz=50.0, seed_noise=41)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 25.0), -127, 127).astype(np.int8)
    cap, win = 12 * n, 2 * n
    kw = dict(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, samples_per_chip=4, samples_per_code=float(n))
    g = StreamGroup.local(devices, cap, win, mode="broadcast")
    g.push(x8, len(x8), "ibyte")
    priv = [SampleStream(cap, win, device=d) for d in devices]
    for r in priv:
        r.push(x8, "ibyte")
    found = {}
    for i, d in enumerate(devices):
        mine = [p for p in range(1, 33) if (p - 1) % n_dev == i]
        res = []
        for ring in (g.ring(i), priv[i]):
            acq = PcpsAcquisitionBank(max_prn=len(mine), device=d, **kw)
            for k, p in enumerate(mine):
                acq.set_local_code(k, oracle.ca_code_complex_sampled(p, int(fs)))
            res.append(acq.dwell_ring(ring, 2 * n + 123, len(mine)))
            acq.close()
        for k, p in enumerate(mine):
            a, b = res[0][k], res[1][k]
            assert (a["index_time"], a["index_doppler"], a["test_statistics"], a["doppler_hz"]) == (b["index_time"], b["index_doppler"], b["test_statistics"], b["doppler_hz"]), (d, p, a, b)
            found[p] = a
    g.close()
    for p, fd in zip(sig_prns, dops):
        assert abs(found[p]["doppler_hz"] - fd) <= 250 and found[p]["test_statistics"] > 4.0 * np.median([found[q]["test_statistics"] for q in found if q not in sig_prns]), (p, found[p])


@pytest.mark.parametrize("dist_mode", ["broadcast", "scatter_allgather"])
def test_bench_gpus_2_spawns_its_ranks_and_shards(devices, dist_mode):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts its two ranks itself (torch.distributed.run), every rank takes its channels of the
    stream that the engine's group replicates (gsh_stream_group_create_rank), rank 0 prints the line.  On a one-GPU box: both ranks on GPU 0
    (GSH_BENCH_SHARE_GPU), torch.distributed over gloo for the barrier / MAX-reduce / id hand-over, the stand-in library under the engine."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    stub = "GSH_RCCL_LIBRARY" in os.environ
    if stub:
        env.update(GSH_BENCH_SHARE_GPU="1", GSH_BENCH_BACKEND="gloo")
    env["GSH_BENCH_DIST"] = dist_mode
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--settle-steps", "0", "--blocks-per-step", "3",
                        "--epochs", "40"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["stream_group_mode"] == dist_mode, d
    assert d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 2
    assert d["spot_check"]["ranks_checked"] == 2 and d["spot_check"]["worst_err"] <= 1e-6, d["spot_check"]
    assert d["rccl_library_is_test_stub"] == stub
    for leg in ("closed_loop_sharded", "acquisition_sharded"):
        assert leg in d and "error" not in d[leg] and d[leg]["n_gpus"] == 2 and d[leg]["value"] > 0, d.get(leg)


def test_bench_gpus_more_than_visible_fails_loudly(gpu, monkeypatch):
    n = _n_devices()
    monkeypatch.delenv("GSH_BENCH_SHARE_GPU", raising=False)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
