"""CPU models of two round-4 rewrites inside the closed-loop kernel (csrc/tracking_loop.hip, csrc/mcorr_device.h) -- the arguments the code comments make, checked in float32 numpy:

* loop_filter_apply keeps Tracking_loop_filter's two 4-deep histories newest first (a shift register) instead of as a ring with a moving index (T/tracking_loop_filter.cc:63-98):
  the same products added in the same order, so the outputs must be bit-identical for every filter order;
* run_segment_packed (MRG, one seed per window): lanes 0..2 of a wave evaluate the three wave-uniform rotations instead of their own seed phasors and take their seeds from lane 3's,
  two samples back per lane (products with conj(inc)^2): their seeds must agree with the directly evaluated ones to float32 round-off."""
import numpy as np
import pytest

f32 = np.float32


def _ring_filter(in_c, out_c, xs):
    """Tracking_loop_filter::apply as the reference writes it: d_current_index moves, inputs / outputs are read at (index + i) % 4"""
    in_h, out_h, idx, ys = [f32(0)] * 4, [f32(0)] * 4, 3, []
    for x in xs:
        r = f32(0)
        for i, c in enumerate(out_c):
            r = f32(r + f32(c * out_h[(idx + i) % 4]))
        idx = (idx + 3) % 4
        in_h[idx] = f32(x)
        for i, c in enumerate(in_c):
            r = f32(r + f32(c * in_h[(idx + i) % 4]))
        out_h[idx] = r
        ys.append(r)
    return ys


def _shift_filter(in_c, out_c, xs):
    in_h, out_h, ys = [f32(0)] * 4, [f32(0)] * 4, []
    for x in xs:
        r = f32(0)
        for i, c in enumerate(out_c):
            r = f32(r + f32(c * out_h[i]))
        ih = [f32(x), in_h[0], in_h[1], in_h[2]]
        for i, c in enumerate(in_c):
            r = f32(r + f32(c * ih[i]))
        in_h, out_h = ih, [r, out_h[0], out_h[1], out_h[2]]
        ys.append(r)
    return ys


@pytest.mark.parametrize("n_in,n_out", [(1, 0), (2, 1), (3, 2), (4, 3), (2, 0), (3, 1)])
def test_shift_register_histories_equal_the_ring(n_in, n_out):
    rng = np.random.default_rng(7 * n_in + n_out)
    in_c = [f32(v) for v in rng.standard_normal(n_in)]
    out_c = [f32(v) for v in (rng.standard_normal(n_out) * 0.4)]
    xs = rng.standard_normal(300).astype(np.float32)
    a, b = _ring_filter(in_c, out_c, xs), _shift_filter(in_c, out_c, xs)
    assert np.array(a, dtype=np.float32).tobytes() == np.array(b, dtype=np.float32).tobytes()


def _cmul(a, b):
    return (f32(f32(a[0] * b[0]) - f32(a[1] * b[1])), f32(f32(a[0] * b[1]) + f32(a[1] * b[0])))


@pytest.mark.parametrize("step", [1.3e-3, -4.1e-4, 2.5e-2, 0.0])
def test_seeds_of_the_three_lanes_that_evaluate_the_rotations(step):
    rem, n_first, wave = 0.7, -1, 5
    def e(ph):
        return (f32(np.cos(ph)), f32(-np.sin(ph)))  # exp(-j ph), evaluated in double and rounded once: what expmj delivers
    inc = e(step)
    tid3 = 64 * wave + 3
    s3 = e(rem + (n_first + 2 * tid3) * step)
    q = _cmul((inc[0], f32(-inc[1])), (inc[0], f32(-inc[1])))  # conj(inc)^2 = exp(+j 2 step)
    s2 = _cmul(s3, q)
    s1 = _cmul(s2, q)
    s0 = _cmul(s1, q)
    for lane, got in ((2, s2), (1, s1), (0, s0)):
        want = e(rem + (n_first + 2 * (64 * wave + lane)) * step)
        assert abs(float(got[0]) - float(want[0])) < 6e-7 and abs(float(got[1]) - float(want[1])) < 6e-7, (lane, got, want)


def test_thread_0_never_misses_a_loss_of_lock():
    """The lanes of the loop arithmetic meet through LDS words, and thread 0 waits for the lock detectors' verdict only in a period in which a fail counter CAN pass its
    limit: when the detector lane said so at the end of the last period (may_trip = counter + 1 > limit), or when thread 0 itself sees the bit-synchronisation time limit force
    the carrier counter to 300000 (trk.cc:2000-2007).  The word may already hold the value for the NEXT period when thread 0 reads it (the lane has finished): then the verdict
    is in as well, and a limit passed in this period leaves the counter within one of it.  Model: random walks of a counter (trk.cc:1196-1207: +1 on a failed test, -1 on a
    passed one while positive, untouched during the pull-in transitory, cleared once when it ends) -- a loss must imply a wait under either reading of the word."""
    rng = np.random.default_rng(11)
    for trial in range(4000):
        limit = int(rng.integers(0, 6))
        counter, latched = int(rng.integers(0, limit + 2)), True
        may_trip_prev = counter + 1 > limit
        for period in range(60):
            pull_in = period < 5 and trial % 3 == 0
            forced = rng.random() < 0.02
            if latched and not pull_in:
                latched, counter = False, 0
            if forced:
                counter = 300000
            if not pull_in:
                if rng.random() < 0.55:
                    counter += 1
                elif counter > 0:
                    counter -= 1
            lost = counter > limit
            may_trip_next = counter + 1 > limit
            for word in (may_trip_prev, may_trip_next):      # what thread 0 may read: last period's value, or -- the lane being done -- the next one's
                waits = word or forced
                assert waits or not lost, (trial, period, limit, counter)
            if lost:
                break
            may_trip_prev = may_trip_next
