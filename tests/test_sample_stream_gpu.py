"""GPU tests of the device-resident IF sample ring (gsh_stream_*) and of the on-device sample-format conversions.

Conversions restate the reference's data_type_adapter blocks (ibyte_to_complex.cc:45-51, ishort_to_complex.cc:45-51: GNU Radio
interleaved_char/short_to_complex = integer -> float casts, no scaling; inverted_spectrum -> conjugate_cc) and
volk_gnsssdr_16ic_convert_32fc: exact, so the bar is bit equality with numpy's cast.  The ring is checked against a plain
numpy model of "the last C samples of everything pushed so far", and a correlator bank bound to it must return exactly what it
returns on the same samples in a flat buffer."""
import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream, tracking_params_for

pytestmark = pytest.mark.gpu


def _stream(gpu, cap, win):
    from gnss_sdr_amd.sample_stream import SampleStream
    return SampleStream(cap, win, device=gpu)


def _as_complex(a):
    a = np.asarray(a)
    return (a[:, 0].astype(np.float32) + 1j * a[:, 1].astype(np.float32)).astype(np.complex64)


@pytest.mark.parametrize("item,dtype,lo,hi", [("ibyte", np.int8, -128, 128), ("ishort", np.int16, -32768, 32768)])
@pytest.mark.parametrize("inv", [False, True])
def test_integer_items_convert_exactly(gpu, item, dtype, lo, hi, inv):
    rng = np.random.default_rng(11)
    s = _stream(gpu, 1 << 16, 1 << 10)
    model = np.zeros(0, np.complex64)
    # ragged push sizes: vector body + scalar head / tail, odd source alignments inside the staging buffer, empty push
    for n in (1, 2, 3, 4, 5, 1023, 4096, 0, 7777, 31):
        a = rng.integers(lo, hi, size=(n, 2)).astype(dtype)
        if n >= 2:
            a[:2] = [[lo, hi - 1], [hi - 1, lo]]  # extremes
        first = s.push(a, item, inverted_spectrum=inv)
        assert first == len(model)
        c = _as_complex(a)
        model = np.concatenate([model, np.conj(c) if inv else c])
    lo_i, hi_i = s.range()
    assert (lo_i, hi_i) == (0, len(model))
    got = s.read(0, len(model))
    assert np.array_equal(got.view(np.uint32), model.view(np.uint32))  # bit equality, including the sign of conj's zeros


def test_page_locked_items_reach_the_ring_by_dma(gpu):
    """gsh_stream_push_pinned: gr_complex items go by DMA straight into their ring positions (two pieces when the push wraps; the mirror behind the ring's
    end follows: the tracking adapters' GPU tests read their windows through it), inverted-spectrum and integer items through the device staging buffer and
    the conversion kernel.  Model: the last `cap` samples pushed, compared bit for bit after every push."""
    rng = np.random.default_rng(17)
    cap, win = 6000, 1500
    s = _stream(gpu, cap, win)
    model = np.zeros(0, np.complex64)
    plan = [(1500, "gr_complex", False), (4000, "gr_complex", False), (999, "gr_complex", False), (5999, "gr_complex", False), (0, "gr_complex", False),
            (1, "gr_complex", False), (3333, "gr_complex", True), (6000, "gr_complex", False), (777, "ibyte", False), (2048, "ishort", True), (4100, "gr_complex", False)]
    for n, item, inv in plan:
        if item == "gr_complex":
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            c = x
        else:
            dt = np.int8 if item == "ibyte" else np.int16
            x = rng.integers(-100, 100, size=(n, 2)).astype(dt)
            c = _as_complex(x)
        first = s.push_pinned(x, item, inverted_spectrum=inv)
        assert first == len(model)
        model = np.concatenate([model, np.conj(c) if inv else c])
        lo, hi = s.range()
        assert hi == len(model) and lo == max(0, len(model) - cap)
        assert np.array_equal(s.read(lo, hi - lo).view(np.uint32), model[lo:hi].view(np.uint32))
    s.close()


def test_ring_wraps_and_keeps_the_last_capacity_samples(gpu):
    rng = np.random.default_rng(5)
    cap, win = 5000, 1200
    s = _stream(gpu, cap, win)
    model = np.zeros(0, np.complex64)
    for n in (1200, 3000, 999, 2500, 4999, 1, 5000, 123):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        s.push(x)
        model = np.concatenate([model, x])
        lo, hi = s.range()
        assert hi == len(model) and lo == max(0, len(model) - cap)
        assert np.array_equal(s.read(lo, hi - lo), model[lo:hi])
    from gnss_sdr_amd import GshError
    lo, hi = s.range()
    with pytest.raises(GshError):
        s.read(lo - 1, 10)      # already overwritten
    with pytest.raises(GshError):
        s.read(hi - 5, 10)      # not pushed yet
    with pytest.raises(GshError):
        s.push(np.zeros(cap + 2, np.complex64))  # larger than the ring


def test_bank_bound_to_ring_matches_flat_buffer(gpu):
    """Windows addressed by absolute sample index inside the ring -- including windows that straddle the wrap point and
    therefore run into the mirror -- give bit-identical correlator outputs to the same windows of a flat device buffer."""
    from gnss_sdr_amd.tracking import CorrelatorBank
    from gnss_sdr_amd import GshError
    fs, n = 4e6, 4000
    total = 40 * n
    x = synth_gps_l1_stream(total, fs, [1, 2, 3], [1000.0, -2000.0, 300.0], [5.0, 300.0, 800.0], seed_noise=8)
    x8 = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 30.0), -127, 127).astype(np.int8)   # what an 8-bit front-end delivers
    xf = _as_complex(x8)
    cap = 9 * n + 2                                   # not a multiple of the window: every wrap position occurs
    ring = _stream(gpu, cap, 2 * n)
    bank_r, bank_f = CorrelatorBank(3, 1023, device=gpu), CorrelatorBank(3, 1023, device=gpu)
    for c in range(3):
        bank_r.set_code(c, oracle.ca_code(c + 1))
        bank_f.set_code(c, oracle.ca_code(c + 1))
    bank_f.set_stream_host(xf)
    bank_r.set_stream_ring(ring)
    rng = np.random.default_rng(2)
    params = [tracking_params_for(fs, d, rng) for d in (1000.0, -2000.0, 300.0)]
    pushed = 0
    n_wrapping = 0
    for blk in range(0, total, 3 * n + 17):
        m = min(3 * n + 17, total - blk)
        first = ring.push(x8[blk:blk + m], "ibyte")
        assert first == pushed
        pushed += m
        lo, hi = ring.range()
        # every channel correlates a few windows of what is resident, newest first
        jobs = []
        for c in range(3):
            for k in range(3):
                off = hi - n - k * (n // 2 + 3) - c
                if off >= lo:
                    jobs.append(dict(sample_offset=off, n_samples=n, code_slot=c, shifts_chips=[-0.5, 0.0, 0.5], **params[c]))
                    if off % cap + n > cap:
                        n_wrapping += 1
        if not jobs:
            continue
        a = bank_r.correlate(jobs)
        b = bank_f.correlate(jobs)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert n_wrapping >= 3
    lo, hi = ring.range()
    with pytest.raises(GshError):
        bank_r.correlate([dict(sample_offset=lo - 1, n_samples=n, code_slot=0, shifts_chips=[0.0], **params[0])])
    with pytest.raises(GshError):
        bank_r.correlate([dict(sample_offset=hi - n + 1, n_samples=n, code_slot=0, shifts_chips=[0.0], **params[0])])
    with pytest.raises(GshError):
        bank_r.correlate([dict(sample_offset=lo, n_samples=2 * n + 1, code_slot=0, shifts_chips=[0.0], **params[0])])


def test_convert_samples_device_and_push_device(gpu):
    torch = pytest.importorskip("torch")
    from gnss_sdr_amd.sample_stream import convert_samples_device
    dev = torch.device("cuda", gpu)
    g = torch.Generator(device="cpu")
    g.manual_seed(3)
    n = 100003
    raw = torch.randint(-128, 128, (n, 2), dtype=torch.int8, generator=g)
    d_raw = raw.to(dev)
    out = torch.empty(n, dtype=torch.complex64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    convert_samples_device(gpu, d_raw.data_ptr(), "ibyte", out.data_ptr(), n, hip_stream=st)
    torch.cuda.synchronize()
    exp = _as_complex(raw.numpy())
    assert np.array_equal(out.cpu().numpy(), exp)
    s = _stream(gpu, 1 << 18, 1 << 12)
    first = s.push_device(d_raw.data_ptr(), n, "ibyte", inverted_spectrum=True, hip_stream=st)
    torch.cuda.synchronize()
    assert first == 0
    assert np.array_equal(s.read(0, n), np.conj(exp))
