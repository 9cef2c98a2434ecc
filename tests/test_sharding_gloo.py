"""CPU tests of the multi-GPU layout with world_size 2 over gloo: the ingest rank broadcasts the sample block, every
rank correlates its own channel shard, and the union equals the single-process result.  The per-rank arithmetic is
done by the oracle here (no GPU in this container); on the GPU box the same job tables go to gsh_bank_*."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_channels, epochs, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    from gnss_sdr_amd.sharding import broadcast_block, epoch_major_jobs, shard_range
    from helpers import oracle_job, synth_gps_l1_stream, tracking_params_for
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fs = 4e6
    total = (epochs + 2) * n
    if rank == 0:
        x = synth_gps_l1_stream(total, fs, [1, 2], [500.0, -1500.0], [10.0, 700.0], seed_noise=77)
        block = torch.from_numpy(x.view(np.float32).copy())
    else:
        block = torch.zeros(2 * total, dtype=torch.float32)
    broadcast_block(block, src=0)
    x = block.numpy().view(np.complex64)
    rng = np.random.default_rng(123)  # same table on every rank
    per_channel = {c: (int(rng.integers(0, n)), tracking_params_for(fs, float(rng.uniform(-5000, 5000)), rng)) for c in range(n_channels)}
    mine = list(shard_range(n_channels, world, rank))
    rows = epoch_major_jobs(mine, per_channel, epochs, n, [-0.5, 0.0, 0.5])
    out = {}
    for r in rows:
        ch = mine[r["code_slot"]]
        o32, _, _ = oracle_job(oracle.ca_code(ch % 32 + 1), x, r)
        out[(ch, r["sample_offset"])] = o32.copy()
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_helpers():
    from gnss_sdr_amd.sharding import owner_of, shard_range
    for n, w in [(32, 8), (256, 8), (5, 2), (3, 4), (50, 8), (1, 1)]:
        seen = []
        for r in range(w):
            rr = list(shard_range(n, w, r))
            seen += rr
            assert all(owner_of(u, n, w) == r for u in rr)
        assert seen == list(range(n))
        sizes = [len(shard_range(n, w, r)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_broadcast_and_channel_shards():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from gnss_sdr_amd.sharding import epoch_major_jobs
    from helpers import oracle_job, synth_gps_l1_stream, tracking_params_for
    world, n_channels, epochs, n = 2, 5, 2, 4000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_channels, epochs, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = {}
    for g in gathered:
        assert not (set(g) & set(merged))  # shards are disjoint
        merged.update(g)
    # single-process reference over all channels
    fs = 4e6
    x = synth_gps_l1_stream((epochs + 2) * n, fs, [1, 2], [500.0, -1500.0], [10.0, 700.0], seed_noise=77)
    rng = np.random.default_rng(123)
    per_channel = {c: (int(rng.integers(0, n)), tracking_params_for(fs, float(rng.uniform(-5000, 5000)), rng)) for c in range(n_channels)}
    rows = epoch_major_jobs(list(range(n_channels)), per_channel, epochs, n, [-0.5, 0.0, 0.5])
    assert len(merged) == len(rows) == n_channels * epochs
    for r in rows:
        o32, _, _ = oracle_job(oracle.ca_code(r["code_slot"] % 32 + 1), x, r)
        got = merged[(r["code_slot"], r["sample_offset"])]
        assert np.array_equal(got.view(np.float32), o32.view(np.float32))


def _dist_worker(rank, world, port, nbytes, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from gnss_sdr_amd.sharding import BlockDistributor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = BlockDistributor(nbytes, world, rank, src=0, mode=mode)
    g = torch.Generator().manual_seed(5)
    ok = True
    for blk in range(3):  # double-buffered like bench.py: block k+1 travels while block k is in use
        src = None
        if rank == 0:
            src = torch.randint(-128, 128, (d.padded,), dtype=torch.int8, generator=g)
        dst = torch.zeros(d.padded, dtype=torch.int8)
        piece = torch.zeros(d.chunk, dtype=torch.int8)
        works = d.start(dst, src, piece)
        d.finish(works)
        ref = torch.randint(-128, 128, (d.padded,), dtype=torch.int8, generator=torch.Generator().manual_seed(5)) if blk == 0 else None
        if blk == 0:
            ok = ok and bool(torch.equal(dst, ref))
        chk = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(chk, dst.to(torch.int64).sum().reshape(1))
        ok = ok and all(int(c) == int(chk[0]) for c in chk)
    if rank == 0:
        q.put(ok)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    dist.barrier()
    dist.destroy_process_group()
    assert all(flags)


@pytest.mark.parametrize("mode", ["scatter_allgather", "broadcast"])
def test_block_distributor_two_ranks(mode):
    """The raw-sample block reaches every rank intact through scatter + all-gather (and through the plain broadcast fallback);
    nbytes deliberately not a multiple of the world size."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, 100001, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get(timeout=180) is True
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
