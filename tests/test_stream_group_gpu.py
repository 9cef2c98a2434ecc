"""GPU tests of the block-replication group under the C ABI (gsh_stream_group_*, csrc/stream_group.hip) and of the asynchronous ingest
(gsh_stream_push_async / gsh_stream_wait / gsh_stream_seek).

The reference has ONE input buffer that every channel block reads (gnss_flowgraph.cc:1227-1231); the group is that buffer kept
identical in the HBM of several GPUs.  A single MI355X can only form a group of one (RCCL across devices runs on the driver's 8-GPU
node), which still exercises everything but the wire: staging, the cast on the group's stream, the event ordering against readers
and the ring bookkeeping.  Bar: ring contents bit-identical to a local gsh_stream_push of the same items, and a correlator bank bound
to the group's ring returns exactly what it returns on a private ring."""
import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream, tracking_params_for

pytestmark = pytest.mark.gpu


def _blocks(rng, sizes):
    return [rng.integers(-128, 128, size=(n, 2)).astype(np.int8) for n in sizes]


@pytest.mark.parametrize("rccl", [False, True], ids=["no_exchange", "one_rank_rccl"])
@pytest.mark.parametrize("mode", ["broadcast", "scatter_allgather"])
@pytest.mark.parametrize("how", ["local", "rank", "rank_own_id"])
def test_group_of_one_equals_local_push(gpu, mode, how, rccl):
    """rccl = True (GSH_GROUP_FORCE_RCCL): the group of one builds a one-rank communicator (ncclCommInitAll / ncclCommInitRank) and every block goes
    through ncclBroadcast, or the grouped ncclSend / ncclRecv to itself + ncclAllGather, on the ring's stream -- the only RCCL evidence a single-GPU
    box can give: the dlopen, the hand-declared signatures, the ncclUniqueId by value, the stream ordering against the cast that follows."""
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    cap, win = 40000, 9000
    if how == "local":
        g = StreamGroup.local([gpu], cap, win, mode=mode, force_rccl=rccl)
    elif how == "rank":
        g = StreamGroup.from_rank(gpu, 0, 1, StreamGroup.unique_id(), cap, win, mode=mode, force_rccl=rccl)
    else:
        g = StreamGroup.from_rank(gpu, 0, 1, None, cap, win, mode=mode, force_rccl=rccl)
    assert g.size() == 1
    info = g.rccl_info()
    assert info["ranks"] == (1 if rccl else 0) and info["collectives"] == 0
    if rccl:
        assert info["version"] > 20000, info    # ncclGetVersion answered: RCCL is loaded and its symbols resolve
    ring = g.ring(0)
    ref = SampleStream(cap, win, device=gpu)
    rng = np.random.default_rng(3)
    total = 0
    for a in _blocks(rng, (9000, 1, 8191, 20000, 0, 777, 33333, 4096)):   # ragged, empty, larger than half the ring, wraps
        first = g.push(a, len(a), "ibyte")
        assert first == total == ref.push(a, "ibyte")
        total += len(a)
        g.wait()
        lo, hi = ring.range()
        assert (lo, hi) == ref.range() == (max(0, total - cap), total)
        n = min(hi - lo, win)
        for start in (lo, hi - n):
            assert np.array_equal(ring.read(start, n).view(np.uint32), ref.read(start, n).view(np.uint32))
    info = g.rccl_info()
    # 7 non-empty blocks: one broadcast each, or send + receive + all-gather each
    assert info["collectives"] == (0 if not rccl else 7 * (1 if mode == "broadcast" else 3)), info
    if rccl:
        print(f"one-rank RCCL group ({how}, {mode}): ncclGetVersion {info['version']}, {info['collectives']} RCCL calls, ring bit-identical to a local push")
    g.close()


@pytest.mark.parametrize("rccl", [None, "broadcast", "scatter_allgather"], ids=["no_exchange", "rccl_broadcast", "rccl_scatter_allgather"])
def test_bank_on_group_ring_is_bit_identical_to_private_ring(gpu, rccl):
    from gnss_sdr_amd.sample_stream import SampleStream, StreamGroup
    from gnss_sdr_amd.tracking import CorrelatorBank
    fs, n = 4e6, 4000
    total = 30 * n
    dopplers = [1000.0, -2000.0, 300.0]
    x = synth_gps_l1_stream(total, fs, [1, 2, 3], dopplers, [5.0, 300.0, 800.0], seed_noise=21)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 30.0), -127, 127).astype(np.int8)
    cap = 9 * n + 2
    # (rccl: the block reaches the ring through a one-rank communicator's collectives; the launch that reads it is ordered behind them by the ring's events alone)
    g = StreamGroup.local([gpu], cap, 2 * n, mode=rccl or "broadcast", force_rccl=rccl is not None)
    priv = SampleStream(cap, 2 * n, device=gpu)
    rng = np.random.default_rng(2)
    params = [tracking_params_for(fs, d, rng) for d in dopplers]
    outs = []
    for ring, push in ((g.ring(0), lambda a: g.push(a, len(a), "ibyte")), (priv, lambda a: priv.push(a, "ibyte"))):
        bank = CorrelatorBank(3, 1023, device=gpu)
        for c in range(3):
            bank.set_code(c, oracle.ca_code(c + 1))
        bank.set_stream_ring(ring)
        got = []
        for blk in range(0, total, 3 * n + 17):                       # no host wait between the group's push and the launch that reads it
            m = min(3 * n + 17, total - blk)
            push(x8[blk:blk + m])
            lo, hi = ring.range()
            jobs = [dict(sample_offset=hi - n - k * (n // 2 + 3) - c, n_samples=n, code_slot=c, shifts_chips=[-0.5, 0.0, 0.5], **params[c])
                    for c in range(3) for k in range(3) if hi - n - k * (n // 2 + 3) - c >= lo]
            if jobs:
                got.append(bank.correlate(jobs))
        outs.append(np.concatenate(got, axis=0))
        bank.close()
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert (g.rccl_info()["collectives"] > 0) == (rccl is not None)
    g.close()


def test_push_async_equals_push_and_orders_against_readers(gpu):
    from gnss_sdr_amd.sample_stream import SampleStream
    cap, win = 30000, 6000
    a_sync, a_async = SampleStream(cap, win, device=gpu), SampleStream(cap, win, device=gpu)
    rng = np.random.default_rng(8)
    total = 0
    for a in _blocks(rng, (6000, 6000, 12000, 5, 29999, 6000, 6000, 6000)):
        assert a_sync.push(a, "ibyte") == a_async.push_async(a, "ibyte") == total
        total += len(a)
    a_async.wait()
    lo, hi = a_async.range()
    assert (lo, hi) == a_sync.range()
    assert np.array_equal(a_async.read(hi - win, win).view(np.uint32), a_sync.read(hi - win, win).view(np.uint32))
    # seek: the next push lands at the absolute index given, nothing older is resident (a channel re-started after a gap in the stream)
    a_async.seek(10_000_000)
    a = _blocks(rng, (4000,))[0]
    assert a_async.push_async(a, "ibyte") == 10_000_000
    a_async.wait()
    assert a_async.range() == (10_000_000, 10_004_000)
    c = (a[:, 0].astype(np.float32) + 1j * a[:, 1].astype(np.float32)).astype(np.complex64)
    assert np.array_equal(a_async.read(10_000_000, 4000), c)
    from gnss_sdr_amd import GshError
    with pytest.raises(GshError):
        a_async.read(10_000_000 - 10, 20)
