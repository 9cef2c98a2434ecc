"""CPU models of two pieces of the batched correlator's paired trip (gnss-sdr_amd/csrc/mcorr_device.h, round 6), restated in numpy so that their claims can be
checked without a GPU (the GPU tests hold the kernel itself to the oracle's chips: tests/test_tracking_gpu.py):

  * TWO FLOORS PER INSTRUCTION (packed_trip, GSH_MC_PKRTZ).  The chip-index chain k = floor((step * (float)n + shift) - rem), one IEEE rounding per operation
    (the reference: volk_gnsssdr_32f_xn_resampler_32f_xn.h:75-76), is evaluated on constants scaled by 2^-24 and converted by v_cvt_pkrtz_f16_f32 -- round toward
    zero to half precision -- whose result's BIT PATTERN is claimed to be the integer floor(u) for 1 <= u < 2048.  Two claims: (a) the scaled chain is the unscaled
    chain times 2^-24 bit for bit as long as no scaled constant is denormal; (b) the half-precision pattern of u * 2^-24 under round-toward-zero is floor(u).
  * COUNTED RUNS (run_segment_packed, GSH_MC_RUNLEN).  The number of trips from i on that may pair their taps is read off the judgement masks with shifts and a
    count of trailing ones; it must equal what asking the per-trip flags one trip after the other gives, up to the two places where a run is allowed to end early
    (a mask word's end, the last plain trip)."""
import numpy as np

F32 = np.float32


def f32_to_f16_bits_rtz(x):
    """Half-precision bit pattern of float32 x (finite, |x| < 65520) rounded TOWARD ZERO, as v_cvt_pkrtz_f16_f32 forms each half; denormal results included."""
    x = np.asarray(x, F32)
    b = x.view(np.uint32).astype(np.uint64)
    sign = (b >> 31) & 1
    e = ((b >> 23) & 0xFF).astype(np.int64)          # biased float32 exponent
    m = (b & 0x7FFFFF) | np.where(e > 0, 0x800000, 0).astype(np.uint64)   # 24-bit significand (hidden bit for normals)
    # value = m * 2^(e - 150) (e >= 1; float32 denormals are far below half's range and give 0).  half: quantum 2^-24 below 2^-14, else 2^(E - 25) with E the half's
    # biased exponent.  Truncation = integer shift right.
    E = e - 127 + 15                                  # biased half exponent if normal
    out = np.zeros(x.shape, np.uint64)
    normal = E >= 1
    # normal: 10 fraction bits = top 10 bits below the hidden bit of m, truncated
    out = np.where(normal, (np.clip(E, 0, 31).astype(np.uint64) << 10) | ((m >> 13) & 0x3FF), out)
    # denormal half: pattern = floor(value / 2^-24) = m >> (150 - 24 - e) = m >> (126 - e)
    sh = np.clip(126 - e, 0, 63).astype(np.uint64)
    out = np.where(~normal, np.where(126 - e < 64, m >> sh, 0), out)
    return (out | (sign << 15)).astype(np.uint32)


def test_rtz_half_conversion_model_against_numpy_on_exact_values():
    """The model itself: wherever float32 -> float16 is EXACT every rounding mode agrees, so numpy's round-to-nearest conversion pins the bit layout (denormals, the
    denormal / normal boundary at 2^-14, exponents)."""
    vals = [0.0, 2.0 ** -24, 3 * 2.0 ** -24, 1023 * 2.0 ** -24, 2.0 ** -14, 1025 * 2.0 ** -24, 2047 * 2.0 ** -24, 2.0 ** -13, 0.5, 1.0, 1.5, 1024.0, 65504.0, -2.0 ** -24, -0.75]
    x = np.array(vals, F32)
    assert np.array_equal(f32_to_f16_bits_rtz(x), x.astype(np.float16).view(np.uint16).astype(np.uint32))
    # and truncation where it is not exact: just below the next representable half
    assert f32_to_f16_bits_rtz(F32(5.999 * 2.0 ** -24)) == 5 and f32_to_f16_bits_rtz(F32(1.9999)) == np.float16(1.9990234375).view(np.uint16)


def test_half_pattern_of_scaled_position_is_its_floor():
    rng = np.random.default_rng(7)
    u = np.concatenate([
        rng.uniform(1.0, 2047.99, 200000).astype(F32),
        np.arange(1, 2048, dtype=F32),                                    # exact integers
        np.nextafter(np.arange(2, 2049, dtype=F32), F32(0)),              # the float just below every integer
        np.nextafter(np.arange(1, 2048, dtype=F32), F32(4096)),           # ... and just above
        np.array([1.0, 1.5, 1023.5, 1023.99994, 1024.0, 1024.0001, 2039.9999, 2047.9999], F32)])
    scaled = u * F32(2.0 ** -24)                                          # a power of two: exact
    assert np.array_equal(scaled.astype(np.float64), u.astype(np.float64) * 2.0 ** -24)
    pattern = f32_to_f16_bits_rtz(scaled)
    assert np.array_equal(pattern, np.floor(u.astype(np.float64)).astype(np.uint32))
    # the bound of the judgement (every chain value of a paired trip in [1, 2040)) is what keeps this true: at 2048 the quantum doubles
    assert f32_to_f16_bits_rtz(F32(2049.0) * F32(2.0 ** -24)) != 2049


def test_scaled_chain_is_the_unscaled_chain_times_a_power_of_two():
    rng = np.random.default_rng(11)
    n_cases = 4000
    for case in range(n_cases):
        step = F32(rng.choice([1.023e6 / 25e6, 1.023e6 / 4e6, 2.046e6 / 50e6, rng.uniform(0.01, 2.0), rng.integers(8, 2048) / 1024.0]))
        rem = F32(rng.choice([0.0, rng.uniform(-1.0, 2.0), 1e-30, 0.4999999, 0.5]))
        shift = F32(rng.choice([0.0, 0.5, -0.5, 0.25, 0.75, 0.125, 0.3, 1.0]))
        n = np.unique(rng.integers(0, 60000, 64)).astype(F32)
        s = F32(2.0 ** -24)
        ok = all(v == 0 or abs(float(v)) >= 2.0 ** -100 for v in (step, rem, shift))   # scales_exactly() of run_segment_packed
        assert ok
        u = (step * n + shift) - rem                                       # float32, one rounding per operation
        us = ((step * s) * n + (shift * s)) + (-rem * s)
        keep = u >= 1.0                                                    # the judged range: nothing near the denormals
        assert np.array_equal((us[keep].astype(np.float64)) * 2.0 ** 24, u[keep].astype(np.float64)), (case, step, rem, shift)
        in_range = keep & (u < 2040.0)
        assert np.array_equal(f32_to_f16_bits_rtz(us[in_range]), np.floor(u[in_range].astype(np.float64)).astype(np.uint32))


def test_a_tiny_code_phase_is_refused():
    """rem = 1e-33: its scaled value is denormal in float32, the scaled subtraction rounds differently -- scales_exactly() says no and the segment keeps the one-floor form."""
    for rem in (1e-33, -1e-37, 1e-38):
        assert not (F32(rem) == 0 or abs(float(F32(rem))) >= 2.0 ** -100)
    for ok in (0.0, 1e-30, 0.25, -0.3):
        assert F32(ok) == 0 or abs(float(F32(ok))) >= 2.0 ** -100


# ---------------------------------------------------------------------------------------------------------------- counted runs
def flags(i, masks, nch, first_plain, last_plain):
    """mcorr_device.h flags(): bit 0 / 1 = chunk A / B of trip i may pair its taps."""
    nm = len(masks)
    ch = nch * i
    plain = first_plain <= i < last_plain and ch < 64 * nm
    if not plain:
        return 0
    return (masks[ch // 64] >> (ch & 63)) & (3 if nch == 2 else 1)


def paired_run(i0, masks, nch, first_plain, last_plain):
    """mcorr_device.h paired_run(): trips from i0 on that may pair their taps (0: trip i0 may not), from the masks' trailing ones."""
    nm = len(masks)
    ch = nch * i0
    if i0 < first_plain or i0 >= last_plain or ch >= 64 * nm:
        return 0
    m = (masks[ch // 64] >> (ch & 63)) & 0xFFFFFFFFFFFFFFFF
    units = 0x5555555555555555 if nch == 2 else 0xFFFFFFFFFFFFFFFF
    if nch == 2:
        m &= m >> 1
    stop = ~m & units & 0xFFFFFFFFFFFFFFFF
    run = ((stop & -stop).bit_length() - 1) // nch if stop else 64 // nch
    return min(run, last_plain - i0)


def test_counted_runs_agree_with_the_per_trip_question():
    rng = np.random.default_rng(3)
    for case in range(3000):
        nch = int(rng.choice([1, 2]))
        nm = int(rng.choice([2, 4]))
        kind = case % 4
        if kind == 0:
            masks = [int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63) for _ in range(nm)]
        elif kind == 1:   # long runs with a few holes, as real windows have
            masks = []
            for _ in range(nm):
                m = (1 << 64) - 1
                for hole in rng.integers(0, 64, int(rng.integers(0, 4))):
                    m &= ~(1 << int(hole))
                masks.append(m)
        elif kind == 2:
            masks = [(1 << 64) - 1] * nm
        else:
            masks = [0] * nm
        n_trips = int(rng.integers(1, 64 * nm // nch + 8))
        first_plain = int(rng.integers(0, 2))
        last_plain = int(rng.integers(0, n_trips + 1))
        full = 3 if nch == 2 else 1
        # walk the segment as the kernel does: a run of paired trips, then ONE trip of the other kind
        i, paired = 0, []
        while i < n_trips:
            run = paired_run(i, masks, nch, first_plain, last_plain)
            # a run never claims a trip the per-trip question refuses ...
            for k in range(run):
                assert flags(i + k, masks, nch, first_plain, last_plain) == full, (case, i, k)
            # ... and it stops early only at a mask word's end or at the last plain trip
            if flags(i + run, masks, nch, first_plain, last_plain) == full and run > 0:
                assert (nch * (i + run)) % 64 == 0 or i + run == last_plain, (case, i, run)
            if run == 0:
                assert flags(i, masks, nch, first_plain, last_plain) != full
            paired += list(range(i, i + run))
            i += run
            if i >= n_trips:
                break
            i += 1   # the trip of the other kind (always correct, only slower)
        assert all(flags(t, masks, nch, first_plain, last_plain) == full for t in paired)
        assert i >= n_trips and (not paired or max(paired) < min(last_plain, n_trips))
