"""CPU coverage of the N > 1 path THAT SHIPS, world size 2 and 3 over gloo -- no GPU in this container:

  * the exchange plan of csrc/stream_group.hip -- gsh_stream_group_plan, the very list gsh_stream_group_push walks to issue its RCCL calls: chunk sizes,
    padded tails, offsets, peers, grouping -- is fetched from libgnss_sdr_hip.so on every rank and EXECUTED over torch.distributed / gloo on host tensors
    (ncclBroadcast -> dist.broadcast, grouped ncclSend / ncclRecv -> batched isend / irecv, ncclAllGather -> all_gather_into_tensor); every rank must end
    up with the ingest rank's block, in both group modes, for sizes that are not multiples of the world size, with two blocks in flight on alternating slots;
  * the same plans executed by a memory-only interpreter for worlds 1..8 (single process);
  * gnss-sdr_amd/sharding.py -- the control plane bench.py --gpus N runs on every rank: launcher environment -> ControlPlane, the communicator id made
    by the engine on rank 0 (gsh_comm_unique_id) handed to every rank, barrier, MAX / MIN / SUM reductions, channel -> GPU and PRN -> GPU maps;
  * channel shards correlated per rank (the oracle does the arithmetic here; on the GPU box the same tables go to gsh_bank_*) -- their union equals the
    single-process result.
The GPU side of the same code runs in tests/test_stream_group_multi_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_RCCL = os.path.join(ROOT, "tests", "host", "libfake_rccl.so")
SIZES = (100001, 16, 1, 48000, 65537)   # raw bytes per block: not multiples of the world size, one below a chunk's 16-byte granule


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(target, world, *args, timeout=240):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return dict(results)


def _rank_env(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))


# ------------------------------------------------------------------------------------------ the plan, interpreted in memory
def _run_plans_in_memory(nbytes, world, mode, block):
    """Every rank's plan executed against numpy buffers; returns the staging buffers afterwards."""
    from gnss_sdr_amd.sample_stream import StreamGroup
    plans = [StreamGroup.plan(nbytes, world, r, mode) for r in range(world)]
    padded = plans[0][0]
    assert all(p[0] == padded for p in plans) and padded >= nbytes and padded % world == 0 and (padded // world) % 16 == 0
    assert padded - nbytes < 16 * world + world            # no more padding than the granule asks for
    stage = [np.full(padded, 0x55, np.uint8) for _ in range(world)]
    piece = [np.full(padded // world, 0xAA, np.uint8) for _ in range(world)]
    stage[0][:nbytes] = block
    bufs = lambda r, which: {"stage": stage[r], "piece": piece[r]}[which]
    n_phase = 1 + max(op["phase"] for _, ops in plans for op in ops)
    for phase in range(n_phase):
        ops = [(r, op) for r, (_, ops) in enumerate(plans) for op in ops if op["phase"] == phase]
        sends = [(r, op) for r, op in ops if op["op"] == "send"]
        for r, op in ops:
            if op["op"] == "recv":
                k = next(i for i, (sr, sop) in enumerate(sends) if sr == op["peer"] and sop["peer"] == r)   # oldest matching send
                sr, sop = sends.pop(k)
                assert sop["bytes"] == op["bytes"]
                bufs(r, op["dst_buf"])[op["dst_offset"]:op["dst_offset"] + op["bytes"]] = bufs(sr, sop["src_buf"])[sop["src_offset"]:sop["src_offset"] + sop["bytes"]]
        assert not sends, "a send nobody receives"
        coll = [(r, op) for r, op in ops if op["op"] in ("broadcast", "allgather")]
        if coll:
            assert len(coll) == world and len({(op["op"], op["bytes"], op["peer"] if op["op"] == "broadcast" else 0) for _, op in coll}) == 1
            if coll[0][1]["op"] == "broadcast":
                root = coll[0][1]["peer"]
                rop = dict(coll)[root]
                data = bufs(root, rop["src_buf"])[rop["src_offset"]:rop["src_offset"] + rop["bytes"]].copy()
                for r, op in coll:
                    bufs(r, op["dst_buf"])[op["dst_offset"]:op["dst_offset"] + op["bytes"]] = data
            else:
                parts = [bufs(r, op["src_buf"])[op["src_offset"]:op["src_offset"] + op["bytes"]].copy() for r, op in sorted(coll, key=lambda t: t[0])]
                for r, op in coll:
                    for k, part in enumerate(parts):
                        o = op["dst_offset"] + k * op["bytes"]
                        bufs(r, op["dst_buf"])[o:o + op["bytes"]] = part
    return stage


@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_plan_replicates_the_block_for_every_world(mode):
    rng = np.random.default_rng(8)
    for world in range(1, 9):
        for nbytes in SIZES + (world * 16, world * 16 - 1, world * 16 + 1):
            block = rng.integers(0, 256, nbytes, dtype=np.uint8)
            for r, st in enumerate(_run_plans_in_memory(nbytes, world, mode, block)):
                assert np.array_equal(st[:nbytes], block), (mode, world, nbytes, r)


def test_plan_shape_and_argument_checks():
    from gnss_sdr_amd import _lib
    from gnss_sdr_amd.sample_stream import StreamGroup
    padded, ops = StreamGroup.plan(1000, 4, 0, "scatter_allgather")
    assert padded == 1024 and [o["op"] for o in ops] == ["send"] * 4 + ["recv", "allgather"]
    assert [o["src_offset"] for o in ops[:4]] == [0, 256, 512, 768] and [o["peer"] for o in ops[:4]] == [0, 1, 2, 3]   # one piece per xGMI link, its own to itself
    padded, ops = StreamGroup.plan(1000, 4, 3, "scatter_allgather")
    assert [o["op"] for o in ops] == ["recv", "allgather"] and ops[0]["peer"] == 0 and ops[0]["bytes"] == 256 and ops[1]["phase"] == 1
    padded, ops = StreamGroup.plan(1000, 4, 2, "broadcast")
    assert len(ops) == 1 and ops[0]["op"] == "broadcast" and ops[0]["bytes"] == padded == 1024 and ops[0]["peer"] == 0
    for bad in (dict(world=0, rank=0), dict(world=2, rank=2), dict(world=65, rank=0)):
        with pytest.raises(_lib.GshError):
            StreamGroup.plan(100, bad["world"], bad["rank"], "broadcast")


# ------------------------------------------------------------------------------------------ the plan, executed over gloo
def _plan_worker(rank, world, port, q, mode):
    _rank_env(rank, world, port)
    import torch
    import torch.distributed as dist
    from gnss_sdr_amd.sample_stream import StreamGroup
    from gnss_sdr_amd.sharding import ControlPlane
    cp = ControlPlane("gloo")
    ok = True
    stage = [None, None]
    piece = [None, None]
    slot = 0
    gen = torch.Generator().manual_seed(5)   # every rank draws the same blocks: rank 0 supplies them, the others check against them
    for nbytes in SIZES:
        block = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, generator=gen)
        padded, ops = StreamGroup.plan(nbytes, world, rank, mode)
        chunk = padded // world
        stage[slot] = torch.full((padded,), 0x55, dtype=torch.uint8)
        piece[slot] = torch.full((chunk,), 0xAA, dtype=torch.uint8)
        if rank == 0:
            stage[slot][:nbytes] = block
        buf = {"stage": stage[slot], "piece": piece[slot]}
        for phase in range(1 + max(o["phase"] for o in ops)):
            p2p = []
            for o in (o for o in ops if o["phase"] == phase):
                src = buf[o["src_buf"]][o["src_offset"]:o["src_offset"] + o["bytes"]] if o["src_buf"] else None
                dst = buf[o["dst_buf"]][o["dst_offset"]:o["dst_offset"] + o["bytes"]] if o["dst_buf"] else None
                if o["op"] == "broadcast":
                    assert src.data_ptr() == dst.data_ptr()     # in place, as the engine issues it
                    dist.broadcast(dst, src=o["peer"])
                elif o["op"] == "send":
                    if o["peer"] == rank:                       # (gloo has no send-to-self; RCCL does: the matching recv below takes it straight)
                        self_piece = src
                    else:
                        p2p.append(dist.P2POp(dist.isend, src, o["peer"]))
                elif o["op"] == "recv":
                    if o["peer"] == rank:
                        dst.copy_(self_piece)
                    else:
                        p2p.append(dist.P2POp(dist.irecv, dst, o["peer"]))
                elif o["op"] == "allgather":
                    out = buf[o["dst_buf"]][o["dst_offset"]:o["dst_offset"] + world * o["bytes"]]
                    dist.all_gather_into_tensor(out, src.clone())
            if p2p:                                             # one group per phase, like ncclGroupStart / ncclGroupEnd
                for w in dist.batch_isend_irecv(p2p):
                    w.wait()
        ok = ok and bool(torch.equal(stage[slot][:nbytes], block))
        ok = ok and cp.same_everywhere(float(stage[slot][:nbytes].to(torch.int64).sum()))
        slot ^= 1
    q.put((rank, ok))
    cp.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
def test_engine_plan_executed_over_gloo(mode, world):
    res = _launch(_plan_worker, world, mode)
    assert res == {r: True for r in range(world)}, res


# ------------------------------------------------------------------------------------------ the control plane of bench.py --gpus N
def test_shard_maps():
    from gnss_sdr_amd.sharding import channel_owner, channels_of, prn_owner, prns_of, weak_channel_prn
    for n, w in [(32, 8), (256, 8), (5, 2), (3, 4), (50, 8), (1, 1)]:
        seen = sorted(c for r in range(w) for c in channels_of(r, w, n))
        assert seen == list(range(n))
        assert all(channel_owner(c, w) == r for r in range(w) for c in channels_of(r, w, n))
        sizes = [len(channels_of(r, w, n)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    for w in (1, 2, 3, 4, 8):
        assert sorted(p for r in range(w) for p in prns_of(r, w)) == list(range(1, 33))
        assert all(prn_owner(p, w) == r for r in range(w) for p in prns_of(r, w))
    assert [weak_channel_prn(1, 32, s) for s in (0, 31)] == [1, 32] and weak_channel_prn(0, 50, 49) == 18
    with pytest.raises(ValueError):
        channels_of(2, 2, 4)


def _control_worker(rank, world, port, q, n_channels, epochs, n):
    _rank_env(rank, world, port)
    os.environ["GSH_RCCL_LIBRARY"] = FAKE_RCCL    # gsh_comm_unique_id without a GPU: the stand-in library makes the id (real RCCL wants a device)
    import oracle
    from gnss_sdr_amd.sharding import ControlPlane, channels_of
    from helpers import oracle_job, synth_gps_l1_stream, tracking_params_for
    cp = ControlPlane("gloo")
    uid = cp.communicator_id()
    ok = isinstance(uid, (bytes, bytearray)) and len(uid) == 128 and uid.startswith(b"fake_rccl_")
    ids = [None] * world
    cp.dist.all_gather_object(ids, bytes(uid))
    ok = ok and all(i == ids[0] for i in ids)
    ok = ok and cp.reduce_max(float(rank)) == world - 1 and cp.reduce_min(float(rank)) == 0 and cp.reduce_sum(1.0) == world
    ok = ok and cp.same_everywhere(3.0) and not cp.same_everywhere(float(rank))
    # every rank correlates ITS channels of the same stream
    fs = 4e6
    x = synth_gps_l1_stream((epochs + 2) * n, fs, [1, 2], [500.0, -1500.0], [10.0, 700.0], seed_noise=77)
    rng = np.random.default_rng(123)  # same table on every rank
    per_channel = {c: (int(rng.integers(0, n)), tracking_params_for(fs, float(rng.uniform(-5000, 5000)), rng)) for c in range(n_channels)}
    out = {}
    for c in channels_of(rank, world, n_channels):
        off, p = per_channel[c]
        for e in range(epochs):
            row = dict(sample_offset=off + e * n, n_samples=n, code_slot=0, shifts_chips=[-0.5, 0.0, 0.5], **p)
            out[(c, row["sample_offset"])] = oracle_job(oracle.ca_code(c % 32 + 1), x, row)[0].copy()
    cp.barrier()
    q.put((rank, (ok, out)))
    cp.close()


@pytest.mark.parametrize("world", [2, 3])
def test_control_plane_and_channel_shards(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    assert os.path.exists(FAKE_RCCL), "tests/host/libfake_rccl.so was not built (__graft_entry__.build)"
    import oracle
    from helpers import oracle_job, synth_gps_l1_stream, tracking_params_for
    n_channels, epochs, n = 5, 2, 4000
    res = _launch(_control_worker, world, n_channels, epochs, n)
    merged = {}
    for r in range(world):
        ok, out = res[r]
        assert ok, r
        assert not (set(out) & set(merged))  # shards are disjoint
        merged.update(out)
    fs = 4e6
    x = synth_gps_l1_stream((epochs + 2) * n, fs, [1, 2], [500.0, -1500.0], [10.0, 700.0], seed_noise=77)
    rng = np.random.default_rng(123)
    per_channel = {c: (int(rng.integers(0, n)), tracking_params_for(fs, float(rng.uniform(-5000, 5000)), rng)) for c in range(n_channels)}
    assert len(merged) == n_channels * epochs
    for c in range(n_channels):
        off, p = per_channel[c]
        for e in range(epochs):
            row = dict(sample_offset=off + e * n, n_samples=n, code_slot=0, shifts_chips=[-0.5, 0.0, 0.5], **p)
            o32 = oracle_job(oracle.ca_code(c % 32 + 1), x, row)[0]
            assert np.array_equal(merged[(c, row["sample_offset"])].view(np.float32), o32.view(np.float32))


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py --gpus N spawns its own ranks and fails loudly when the node has fewer GPUs (here: none)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GSH_BENCH_SHARE_GPU")})
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-400:])
