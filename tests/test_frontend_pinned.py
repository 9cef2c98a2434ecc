"""SURVEY.md 8f-3, the parts of the signal conditioner that CAN be pinned: the reference's own direct resamplers (direct_resampler_conditioner_cc / _cb / _cs) and its
data-type adapter blocks (cshort_to_gr_complex, interleaved_byte_to_complex_byte, interleaved_byte_to_complex_short, interleaved_short_to_complex_short), compiled
from where they lie into oracle/_ref/libgnsssdr_ref_filt.so (oracle/Makefile, round 6) and driven through general_work (oracle/ref_filt_api.cc: refconv_*), hold

  * oracle.direct_resampler (the statement-by-statement restatement in oracle/gnss_oracle_loop.c) -- every output sample bit-equal, for every ratio of
    tests/test_resampler.py, however the scheduler cuts the stream into calls, as long as a call sees the input it needs (a scheduler that offers exactly the
    block's forecast can starve an interpolating call by one item: the block then reads one item past its window and re-uses it -- shown below, not reproduced);
  * the engine's device kernels (-m gpu): gsh_direct_resample_device and the casts of gsh_convert_samples_device / the sample ring, bit-equal to the blocks' outputs.

The FIR / frequency-translating FIR filters stay UNPINNED: their arithmetic lives in GNU Radio (gr::filter), which this image does not have (oracle/fir_oracle.py says so)."""
import numpy as np
import pytest

import oracle
from oracle import ref_filt as rf

pytestmark = pytest.mark.skipif(not rf.available(), reason="oracle/_ref/libgnsssdr_ref_filt.so not built (needs /root/reference at build time)")

RATIOS = [(25e6, 4e6), (32e6, 4e6), (50e6, 25e6), (4e6, 4e6), (12.5e6, 2.048e6), (2.048e6, 2.046e6), (5e6, 4999999.0), (3e6, 1e6),
          (4e6, 10e6), (2.046e6, 8.184e6), (1e6, 1.000001e6), (7e6, 2e6)]
CALLS = [(2048,), (1, 7, 512, 3, 1000), (333,)]


@pytest.mark.parametrize("fs_in,fs_out", RATIOS)
def test_restated_resampler_equals_the_reference_block(fs_in, fs_out):
    rng = np.random.default_rng(int(fs_in + 3 * fs_out) % 9973)
    n = 40011
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    one_call = oracle.direct_resampler(x, fs_in, fs_out)
    for calls in CALLS:
        blk = rf.RefConvBlock(rf.K_RESAMPLER_CC, fs_in, fs_out)
        y, consumed = blk.run(x, calls)
        blk.close()
        ragged = oracle.direct_resampler(x, fs_in, fs_out, call_sizes=list(calls))
        m = min(len(y), len(one_call))
        assert m >= 0.6 * len(one_call) and m > 1000, (calls, len(y), len(one_call))
        assert np.array_equal(y[:m].view(np.uint32), one_call[:m].view(np.uint32)), (fs_in, fs_out, calls)
        m = min(len(y), len(ragged))
        assert np.array_equal(y[:m].view(np.uint32), ragged[:m].view(np.uint32)), (fs_in, fs_out, calls)


def test_byte_and_short_resamplers_select_the_same_items_as_the_complex_one():
    """_cb and _cs are the same loop over lv_8sc_t / lv_16sc_t items: item selection identical to _cc's."""
    rng = np.random.default_rng(5)
    n = 30000
    idx = np.arange(n)
    for fs_in, fs_out in ((25e6, 4e6), (4e6, 10e6), (12.5e6, 2.048e6)):
        cc = rf.RefConvBlock(rf.K_RESAMPLER_CC, fs_in, fs_out)
        picked, _ = cc.run((idx + 0j).astype(np.complex64), (1000, 17))
        picked = picked.real.astype(np.int64)                     # which input item every output is (indices are exact in float32 below 2^24)
        exp = oracle.direct_resampler((idx + 0j).astype(np.complex64), fs_in, fs_out).real.astype(np.int64)
        assert np.array_equal(picked, exp[:len(picked)])
        for kind, dt in ((rf.K_RESAMPLER_CB, np.int8), (rf.K_RESAMPLER_CS, np.int16)):
            items = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max + 1, size=(n, 2)).astype(dt)
            b = rf.RefConvBlock(kind, fs_in, fs_out)
            y, _ = b.run(items, (1000, 17))
            assert len(y) == len(picked) and np.array_equal(y, items[picked]), (kind, fs_in, fs_out)


def test_starved_interpolating_call_slips_in_the_reference_only():
    """What is NOT restated: offered exactly its forecast, an interpolating direct_resampler_conditioner_cc can need one item more than it was given
    (forecast: int((n + 1) fs_in / fs_out); the loop advances ceil-ish), reads it past its window, and consume_each(min(count, ninput)) leaves it to be read
    again -- the output repeats an item from there on.  A property of the block under a minimal scheduler; the engine (and the restatement, which stops at
    the edge of what it was given) resample the STREAM, not the calls."""
    n = 20011
    x = (np.arange(n) + 0j).astype(np.complex64)
    b = rf.RefConvBlock(rf.K_RESAMPLER_CC, 4e6, 10e6)
    y, _ = b.run(np.concatenate([x, x[:8]]), (4096,), offer="forecast")
    full = oracle.direct_resampler(x, 4e6, 10e6)
    m = min(len(y), len(full))
    first = int(np.nonzero(y[:m] != full[:m])[0][0])
    assert first % 4096 == 0 and first > 0 and np.all(y[first:m].real <= full[first:m].real)     # slipped back by an item at a call boundary


def _numpy_cast(pairs):
    """integer (I, Q) pairs -> complex64 by the plain integer -> float32 conversion: the yardstick of the engine's device casts (tests/test_sample_stream_gpu.py)"""
    return (pairs[:, 0].astype(np.float32) + 1j * pairs[:, 1].astype(np.float32)).astype(np.complex64)


@pytest.mark.parametrize("kind", ["cshort_to_gr_complex", "ibyte_to_cbyte", "ibyte_to_cshort", "ishort_to_cshort"])
def test_data_type_adapter_blocks_are_the_plain_casts(kind):
    """The reference's adapter blocks (integer -> integer regrouping; int16 -> float32 through volk_gnsssdr_16ic_convert_32fc_generic) ARE the plain integer ->
    float casts the engine's device conversions are held to: bit-equal on random items and on the corners of the integer ranges."""
    rng = np.random.default_rng(11)
    n = 10007
    if kind == "cshort_to_gr_complex":
        items = rng.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
        items[:4] = [[-32768, 32767], [0, -1], [1, 0], [32767, -32768]]
        b = rf.RefConvBlock(rf.K_CSHORT_TO_GR_COMPLEX)
        y, cons = b.work(items, n)
        assert cons == n and np.array_equal(y.view(np.uint32), _numpy_cast(items).view(np.uint32))
        return
    dt = np.int8 if kind.startswith("ibyte") else np.int16
    flat = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max + 1, size=2 * n).astype(dt)
    flat[:4] = [np.iinfo(dt).min, np.iinfo(dt).max, 0, -1]
    b = rf.RefConvBlock({"ibyte_to_cbyte": rf.K_IBYTE_TO_CBYTE, "ibyte_to_cshort": rf.K_IBYTE_TO_CSHORT, "ishort_to_cshort": rf.K_ISHORT_TO_CSHORT}[kind])
    y, cons = b.work(flat, n)
    assert cons == 2 * n and len(y) == n                   # sync_decimator by 2: two interleaved values per complex item
    assert np.array_equal(y.astype(np.int64), flat.reshape(n, 2).astype(np.int64))
    if y.dtype == np.int16:
        # ... and on through cshort_to_gr_complex, as the reference's conditioner chains them: what reaches the correlators
        z, _ = rf.RefConvBlock(rf.K_CSHORT_TO_GR_COMPLEX).work(y, n)
        assert np.array_equal(z.view(np.uint32), _numpy_cast(flat.reshape(n, 2)).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("fs_in,fs_out", RATIOS)
def test_device_resampler_equals_the_reference_block(gpu, fs_in, fs_out):
    torch = pytest.importorskip("torch")
    from gnss_sdr_amd.sample_stream import direct_resample_device
    dev = torch.device("cuda", gpu)
    rng = np.random.default_rng(7)
    n = 120007
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    blk = rf.RefConvBlock(rf.K_RESAMPLER_CC, fs_in, fs_out)
    exp, _ = blk.run(x, (2048, 100, 3))
    d_x = torch.from_numpy(x).to(dev)
    cap = int(n * max(1.0, fs_out / fs_in)) + 16
    d_y = torch.zeros(cap, dtype=torch.complex64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out0, in0 = 0, 0
    for chunk in (1, 4097, 50000, 33, 60000, n):
        hi = min(n, in0 + chunk)
        if hi <= in0:
            break
        n_out, n_cons = direct_resample_device(gpu, d_x.data_ptr() + 8 * in0, in0, hi - in0, fs_in, fs_out, out0, d_y.data_ptr() + 8 * out0, cap - out0, hip_stream=st)
        out0 += n_out
        in0 += n_cons
        if n_cons == 0 and n_out == 0 and hi == n:
            break
    torch.cuda.synchronize()
    got = d_y.cpu().numpy()[:out0]
    m = min(len(got), len(exp))
    assert m >= 0.9 * len(exp) and m > 0.8 * n * min(1.0, fs_out / fs_in)
    assert np.array_equal(got[:m].view(np.uint32), exp[:m].view(np.uint32))


@pytest.mark.gpu
def test_device_casts_equal_the_reference_adapter_blocks(gpu):
    """gsh_convert_samples_device on cshort / ibyte / ishort items against cshort_to_gr_complex and the interleaved -> complex regrouping blocks followed by the
    integer -> float cast: every output sample bit-equal."""
    torch = pytest.importorskip("torch")
    from gnss_sdr_amd.sample_stream import convert_samples_device
    dev = torch.device("cuda", gpu)
    rng = np.random.default_rng(13)
    n = 50021
    st = torch.cuda.current_stream().cuda_stream
    items = rng.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
    ref, _ = rf.RefConvBlock(rf.K_CSHORT_TO_GR_COMPLEX).work(items, n)
    d_in = torch.from_numpy(items).to(dev)
    d_out = torch.zeros(n, dtype=torch.complex64, device=dev)
    convert_samples_device(gpu, d_in.data_ptr(), "cshort", d_out.data_ptr(), n, hip_stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    for item_type, dt, kind in (("ibyte", np.int8, rf.K_IBYTE_TO_CSHORT), ("ishort", np.int16, rf.K_ISHORT_TO_CSHORT)):
        flat = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max + 1, size=2 * n).astype(dt)
        regrouped, _ = rf.RefConvBlock(kind).work(flat, n)                                   # lv_16sc_t items ...
        ref2, _ = rf.RefConvBlock(rf.K_CSHORT_TO_GR_COMPLEX).work(regrouped.astype(np.int16), n)  # ... then cshort_to_gr_complex, as the reference's conditioner chains them
        d_in = torch.from_numpy(flat).to(dev)
        convert_samples_device(gpu, d_in.data_ptr(), item_type, d_out.data_ptr(), n, hip_stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), ref2.view(np.uint32)), item_type
