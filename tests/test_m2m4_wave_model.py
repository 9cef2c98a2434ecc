"""A lane-by-lane model of m2m4_sums_wave (csrc/tracking_loop.hip): the M2M4 estimator's three sums formed by the 64 lanes of a wave with v_add_f32_dpp row_shr:1 steps
-- fifteen per row of sixteen lanes, rows one after the other, the first lane of a row taking the previous row's last prefix over v_readlane.  The claim the kernel rests
on: after the steps lane n-1 holds ((x0 + x1) + x2 ...) + x(n-1), the additions of T/lock_detectors.cc:68-80's one-thread loop in its order, bit for bit, for every
buffer length 1..64 -- although every row always runs all fifteen steps (a settled lane recomputes the same value).  No GPU: float32 arithmetic in numpy."""
import numpy as np
import pytest


def _row_shr1_add(s, x):
    """v_add_f32_dpp s, s, x row_shr:1 (bound_ctrl off): lane l gets s[l-1] + x[l]; the first lane of each row of sixteen has no source and keeps what it holds"""
    out = s.copy()
    for lane in range(64):
        if lane % 16 != 0:
            out[lane] = np.float32(s[lane - 1] + x[lane])
    return out


def _sums_wave(x, n):
    s = x.copy()
    for base in range(0, n, 16):
        if base > 0:
            s = (np.float32(s[base - 1]) + x).astype(np.float32)  # every lane: readlane(base - 1) + its own term
        for _ in range(15):
            s = _row_shr1_add(s, x)
    return s[n - 1]


@pytest.mark.parametrize("n", list(range(1, 65)))
def test_lane_n_minus_1_holds_the_sequential_sum(n):
    rng = np.random.default_rng(100 + n)
    x = np.zeros(64, dtype=np.float32)
    x[:n] = np.abs(rng.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rng.integers(-3, 6, n).astype(np.float32)  # non-negative terms of very different sizes
    seq = np.float32(0.0)
    for i in range(n):
        seq = np.float32(seq + x[i])
    got = _sums_wave(x, n)
    assert got.tobytes() == seq.tobytes(), (n, float(got), float(seq))
