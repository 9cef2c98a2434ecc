"""Pulse blanking (input filter): oracle behaviour on CPU, device kernel against the oracle on the GPU.
Bars: which segments are blanked -- exact; passed samples bit-identical to the input; noise estimate within 1e-6 relative (float64 vs
float32 segment sums); identical result for any partition of the stream into calls."""
import numpy as np
import pytest

from oracle.pulse_blanking_oracle import PulseBlankingOracle


def _stream(n, seed=1, pulses=((5000, 200, 12.0), (20000, 64, 30.0), (40010, 700, 6.0))):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for start, width, amp in pulses:
        if start + width <= n:
            x[start:start + width] += (amp * np.exp(2j * np.pi * 0.013 * np.arange(width))).astype(np.complex64)
    return x


def test_oracle_blanks_the_pulses_and_only_them():
    x = _stream(60000)
    o = PulseBlankingOracle(pfa=1e-3, length=32, n_segments_est=100, n_segments_reset=1000000)
    assert 90.0 < float(o.thres) < 125.0 and o.thres.dtype == np.float32   # chi-squared, 64 degrees of freedom, upper 1e-3 point ~ 112
    y, used = o.general_work(x)
    assert used == (60000 - 1) // 32 * 32 and len(y) == used               # the strict '<' of :66 leaves the last full segment for the next call
    blanked = np.nonzero(np.all(y.reshape(-1, 32) == 0, axis=1))[0]
    expect = set()
    for start, width in ((5000, 200), (20000, 64), (40010, 700)):
        expect |= set(range(start // 32 + 1, (start + width) // 32))       # segments wholly inside a pulse must go
    assert expect <= set(blanked.tolist())
    assert len(blanked) <= len(expect) + 8                                 # plus at most the partly covered edges and a rare noise hit
    keep = np.ones(used // 32, bool)
    keep[blanked] = False
    assert np.array_equal(y.reshape(-1, 32)[keep], x[:used].reshape(-1, 32)[keep])
    assert abs(float(o.noise_power_estimation) - 1.0) < 0.05               # mean |x|^2 / 2 per degree of freedom of unit-variance components


def test_oracle_partition_independence_and_reestimation():
    x = _stream(30000, seed=4)
    a = PulseBlankingOracle(pfa=1e-2, length=16, n_segments_est=50, n_segments_reset=300)
    ya, _ = a.general_work(x)
    b = PulseBlankingOracle(pfa=1e-2, length=16, n_segments_est=50, n_segments_reset=300)
    parts, pos = [], 0
    for chunk in (1000, 17, 4096, 33, 9000, 16, 15, 100000):
        end = min(len(x), pos + chunk)
        y, used = b.general_work(x[pos:end])
        parts.append(y)
        pos += used                                                        # the scheduler re-presents the unconsumed tail
        if end == len(x) and used == 0:
            break
    yb = np.concatenate(parts)
    assert np.array_equal(ya[:len(yb)], yb) and len(ya) - len(yb) <= 16
    assert a.n_segments <= 301 + 50                                        # the counter was reset and the floor re-estimated (:85-88)


@pytest.mark.gpu
@pytest.mark.parametrize("length,est,reset,pfa", [(32, 100, 1000000, 1e-3), (16, 50, 300, 1e-2), (100, 20, 1000, 0.04), (1, 10, 50, 0.05)])
def test_device_matches_oracle(gpu, length, est, reset, pfa):
    import torch
    from gnss_sdr_amd.sample_stream import PulseBlanking
    x = _stream(60000 if length >= 16 else 6000, seed=7)   # the Python oracle walks segment by segment
    o = PulseBlankingOracle(pfa, length, est, reset)
    g = PulseBlanking(pfa, length, est, reset, device=gpu)
    assert abs(g.threshold - float(o.thres)) <= 2e-6 * float(o.thres)
    dev = torch.device("cuda", gpu)
    d_x = torch.from_numpy(x).to(dev)
    d_y = torch.full((len(x),), complex(7.0, 7.0), dtype=torch.complex64, device=dev)
    pos_g, pos_o, outs = 0, 0, []
    for chunk in (10000, 33, 4096, 25000, 7, 100000):
        end = min(len(x), pos_g + chunk)
        used_g = g.process_device(d_x.data_ptr() + 8 * pos_g, end - pos_g, d_y.data_ptr() + 8 * pos_g)
        y, used_o = o.general_work(x[pos_o:end])
        assert used_g == used_o
        outs.append(y)
        pos_g += used_g
        pos_o += used_o
    torch.cuda.synchronize()
    yo = np.concatenate(outs)
    yg = d_y.cpu().numpy()
    assert np.array_equal(yg[:pos_g].view(np.uint32), yo.view(np.uint32))   # same segments blanked, passed samples bit-identical
    assert np.all(yg[pos_g:] == complex(7.0, 7.0))                          # nothing written past the consumed part
    noise, nseg, last = g.state()
    assert nseg == o.n_segments and last == o.last_filtered
    assert abs(noise - float(o.noise_power_estimation)) <= 1e-6 * float(o.noise_power_estimation)
    # in place
    d_z = torch.from_numpy(x).to(dev)
    g2 = PulseBlanking(pfa, length, est, reset, device=gpu)
    used = g2.process_device(d_z.data_ptr(), len(x), d_z.data_ptr())
    o2 = PulseBlankingOracle(pfa, length, est, reset)
    y2, used2 = o2.general_work(x)
    assert used == used2 and np.array_equal(d_z.cpu().numpy()[:used].view(np.uint32), y2.view(np.uint32))
    g.close()
    g2.close()
