"""GPU parity of the notch input filters (gsh_notch_*, csrc/notch_filter.hip) against oracle/notch_oracle.py, which is pinned to the reference's own Notch and
NotchLite blocks (tests/test_notch_oracle_pinned.py).  Bars: the same segments estimated / filtered / passed for every partition of the stream into calls (one
borderline energy decision may differ where the two floor estimates -- float32 FFT there, double-accumulated DFT here -- differ in their last bits: such a run is
compared up to the first differing segment and must be rare); passed samples bit-identical to the input; filtered samples within 1e-4 of the segment's
amplitude (device libm vs numpy in atan2 / sincos, re-association of the recurrence, both amplified by 1 / (1 - p)); noise estimate within 1e-5."""
import numpy as np
import pytest

from oracle.notch_oracle import NotchLiteOracle, NotchOracle
from test_notch_oracle_pinned import interfered_stream

pytestmark = pytest.mark.gpu

CASES = [
    ("notch", dict(pfa=0.001, p_c_factor=0.9, length=32, n_segments_est=100, n_segments_reset=1000000)),
    ("notch", dict(pfa=0.01, p_c_factor=0.8, length=16, n_segments_est=40, n_segments_reset=600)),
    ("notch", dict(pfa=0.001, p_c_factor=0.95, length=64, n_segments_est=30, n_segments_reset=1000000)),
    ("notch", dict(pfa=0.001, p_c_factor=0.9, length=100, n_segments_est=20, n_segments_reset=200)),
    ("lite", dict(p_c_factor=0.9, pfa=0.001, length=32, n_segments_est=100, n_segments_reset=1000000, n_segments_coeff=8)),
    ("lite", dict(p_c_factor=0.85, pfa=0.01, length=16, n_segments_est=40, n_segments_reset=600, n_segments_coeff=1)),
    ("lite", dict(p_c_factor=0.9, pfa=0.001, length=64, n_segments_est=30, n_segments_reset=1000000, n_segments_coeff=3)),
]


@pytest.mark.parametrize("kind,kw", CASES)
def test_device_matches_oracle(gpu, kind, kw):
    import torch
    from gnss_sdr_amd.sample_stream import NotchFilter
    x = interfered_stream(64000, seed=3 if kind == "notch" else 5)
    o = NotchOracle(**kw) if kind == "notch" else NotchLiteOracle(**kw)
    g = NotchFilter(kw["pfa"], kw["p_c_factor"], kw["length"], kw["n_segments_est"], kw["n_segments_reset"], kw.get("n_segments_coeff", 0), device=gpu)
    assert abs(g.threshold - float(o.thres)) <= 2e-6 * float(o.thres)
    dev = torch.device("cuda", gpu)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.full((len(x),), complex(7.0, 7.0), dtype=torch.complex64, device=dev)
    pos, outs = 0, []
    for chunk in (10000, 33, 4096, 25000, 1 + kw["length"], 100000):
        end = min(len(x), pos + chunk)
        if end - pos < 2:
            break
        used_g = g.process_device(d_x.data_ptr() + 8 * pos, end - pos, d_y.data_ptr() + 8 * pos)
        y, used_o = o.general_work(x[pos:end])
        assert used_g == used_o == len(y)
        outs.append(y)
        pos += used_g
    torch.cuda.synchronize()
    yo = np.concatenate(outs)
    yg = d_y.cpu().numpy()
    assert np.all(yg[pos:] == complex(7.0, 7.0))                            # nothing written past the consumed part
    L = kw["length"]
    modes = np.array(o.modes)
    segs_g, segs_o = yg[:pos].reshape(-1, L), yo.reshape(-1, L)
    scale = np.maximum(np.abs(segs_o).max(axis=1, keepdims=True), 1e-6)
    err = np.abs(segs_g - segs_o).max(axis=1) / scale[:, 0]
    bad = np.nonzero(err > 1e-4)[0]
    first_bad = int(bad[0]) if len(bad) else len(modes)
    assert first_bad >= 0.98 * len(modes), (kind, kw, first_bad, len(modes), err[bad[:4]] if len(bad) else None)
    ok = slice(0, first_bad)
    copied = (modes[ok] != 1)
    assert np.array_equal(segs_g[ok][copied].view(np.uint32), segs_o[ok][copied].view(np.uint32))   # passed segments: the input, bit for bit
    assert (modes == 1).sum() > 100
    if first_bad == len(modes):
        st = g.state()
        assert st["n_segments"] == o.n_segments and st["filter_state"] == o.filter_state
        assert abs(st["noise_pow_est"] - float(o.noise_pow_est)) <= 1e-5 * float(o.noise_pow_est)
        if kind == "lite":
            assert st["n_segments_coeff"] == o.n_segments_coeff and abs(st["z0"] - complex(o.z0)) < 1e-5
        if o.filter_state:
            assert abs(st["last_out"] - complex(o.last_out)) <= 1e-4 * max(1.0, abs(complex(o.last_out)))
    g.close()


def test_steady_state_calls_skip_the_floor_and_argument_rules(gpu):
    """after the estimation phase a call forms no spectra (the shadow state decides); a call that unexpectedly needs them is repeated with them"""
    import torch
    from gnss_sdr_amd import GshError
    from gnss_sdr_amd.sample_stream import NotchFilter
    kw = dict(pfa=0.001, p_c_factor=0.9, length=32, n_segments_est=50, n_segments_reset=400)
    x = interfered_stream(64000, seed=9, cw=((1000, 900, 8.0, 0.11), (30000, 9000, 3.0, -0.19)))   # an interferer INSIDE the estimation phase: the filter engages early
    o = NotchOracle(**kw)
    g = NotchFilter(device=gpu, **kw)
    dev = torch.device("cuda", gpu)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.zeros(len(x), dtype=torch.complex64, device=dev)
    pos = 0
    for chunk in (1200, 700, 3000, 8000, 100000):          # the second and third calls start with n_segments < est and the filter engaged / disengaging
        end = min(len(x), pos + chunk)
        used = g.process_device(d_x.data_ptr() + 8 * pos, end - pos, d_y.data_ptr() + 8 * pos)
        y, used_o = o.general_work(x[pos:end])
        assert used == used_o
        got = d_y[pos:pos + used].cpu().numpy()
        scale = max(1.0, float(np.abs(y).max())) if len(y) else 1.0
        assert np.max(np.abs(got - y)) <= 1e-4 * scale if len(y) else True
        pos += used
    st = g.state()
    assert st["n_segments"] == o.n_segments and abs(st["noise_pow_est"] - float(o.noise_pow_est)) <= 1e-5 * float(o.noise_pow_est)
    with pytest.raises(GshError):
        g.process_device(d_x.data_ptr(), 1000, d_x.data_ptr())        # in place is refused
    with pytest.raises(GshError):
        NotchFilter(length=1, device=gpu)
    assert g.process_device(d_x.data_ptr(), 20, d_y.data_ptr()) == 0   # shorter than a segment: nothing consumed
    g.close()
