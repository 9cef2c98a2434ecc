"""The RCCL path of the block-replication group with MORE THAN ONE GPU (gsh_stream_group_*, csrc/stream_group.hip; SURVEY.md 8e: "RCCL broadcast
of the shared input sample block over xGMI").  The round's GPU boxes have one MI355X, where RCCL only ever sees a group of one
(tests/test_stream_group_gpu.py); these tests skip there and run the moment >= 2 devices are visible -- the driver's 8-GPU node -- so that
the first multi-GPU run proves the collective instead of hoping for it:

  * one process driving all GPUs (gsh_stream_group_create, ncclCommInitAll): after every push, every device's ring holds bit for bit what a
    local gsh_stream_push of the same 8-bit items leaves on that device, in both group modes (broadcast / scatter + all-gather);
  * a correlator bank bound to each device's ring returns exactly what a bank on a private ring of that device returns;
  * one process per GPU (gsh_stream_group_create_rank, the layout bench.py --gpus N uses): ranks are fresh interpreters (multiprocessing, spawn), the
    communicator id travels through a file, rank 0 supplies the blocks, every rank checks its own ring against the items."""
import os
import sys
import tempfile

import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream, tracking_params_for

pytestmark = pytest.mark.gpu


def _n_devices() -> int:
    import gnss_sdr_amd
    return int(gnss_sdr_amd.load().gsh_device_count())


def _need_two():
    n = _n_devices()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the multi-GPU RCCL path needs at least two")
    return n


def _blocks(rng, sizes):
    return [rng.integers(-128, 128, size=(n, 2)).astype(np.int8) for n in sizes]


@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_every_ring_of_the_group_equals_a_local_push(gpu, mode):
    n_dev = _need_two()
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    devices = list(range(n_dev))
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


def test_banks_on_every_device_of_the_group_match_private_rings(gpu):
    n_dev = _need_two()
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    from gnss_sdr_amd.tracking import CorrelatorBank
    fs, n = 4e6, 4000
    total = 24 * n
    dopplers = [1000.0, -2000.0, 300.0]
    x = synth_gps_l1_stream(total, fs, [1, 2, 3], dopplers, [5.0, 300.0, 800.0], seed_noise=21)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 30.0), -127, 127).astype(np.int8)
    cap = 9 * n + 2
    devices = list(range(n_dev))
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


def _rank_main(rank, world, id_path, mode, result_path):
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
    g = StreamGroup.from_rank(rank, rank, world, uid, cap, win, mode=mode)
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
def test_one_process_per_gpu_group(gpu, mode):
    n_dev = _need_two()
    import multiprocessing as mp
    world = min(n_dev, 8)
    ctx = mp.get_context("spawn")  # fresh interpreters: one HIP runtime per rank, nothing inherited from the pytest process
    with tempfile.TemporaryDirectory() as tmp:
        id_path, result_path = os.path.join(tmp, "nccl_id"), os.path.join(tmp, "result")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        procs = [ctx.Process(target=_rank_main, args=(r, world, id_path, mode, result_path)) for r in range(world)]
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
