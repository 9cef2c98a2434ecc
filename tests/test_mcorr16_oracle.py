"""CPU tests of the 16-bit correlator family's checker: the C restatement oracle.mcorr16 against the golden vectors minted from the
reference's own Cpu_Multicorrelator_16sc (tests/golden/make_golden_mcorr16.py), against the live reference where oracle/_ref is built, and the algebra
the HIP kernel's parallel form of the saturating sums rests on (a numpy model of csrc/multicorrelator_16sc.hip's Map16)."""
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcorr16.npz")


def golden_cases():
    g = np.load(GOLD)
    for name in g["names"]:
        name = str(name)
        yield name, {k: g[f"{name}__{k}"] for k in ("x", "code", "shifts", "par", "out")}


@pytest.mark.parametrize("name,case", list(golden_cases()), ids=[n for n, _ in golden_cases()])
def test_restatement_equals_the_golden_vectors(name, case):
    p = [float(v) for v in case["par"]]
    got = oracle.mcorr16(case["code"], case["shifts"], case["x"], p[0], p[1], p[2], p[3])
    assert np.array_equal(got, case["out"]), (name, got.tolist(), case["out"].tolist())


def test_restatement_equals_the_live_reference_on_random_calls():
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_mcorr16_run"):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    rng = np.random.default_rng(1616)
    for case in range(150):
        n = int(rng.integers(1, 7000))
        L = int(rng.choice([1023, 2046, 511, 10230]))
        amp = int(rng.choice([1, 3, 40, 300, 3000, 32767]))
        camp = int(rng.choice([1, 1, 1, 7, 181]))
        x = rng.integers(-amp, amp + 1, size=(n, 2)).astype(np.int16)
        code = rng.integers(-camp, camp + 1, size=(L, 2)).astype(np.int16)
        shifts = np.sort(rng.uniform(-1, 1, int(rng.choice([1, 3, 5])))).astype(np.float32)
        args = (code, shifts, x, float(rng.uniform(0, 6.28)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0, 1)), float(rng.uniform(0.01, 0.6)))
        assert np.array_equal(oracle.mcorr16(*args), oracle.ref_mcorr16(*args)), case
    # the phasors the class forms before it calls the kernel: the restatement's C expressions against the reference library's C++ ones
    for rem, step in ((0.0, 0.0), (1.234, 0.0321), (6.2, -0.4), (-3.0, 1e-7), (100.0, 3.0)):
        want = np.empty(4, np.float32)
        R.ref_mcorr16_phasors(rem, step, want)
        assert np.array_equal(oracle.mcorr16_phasors(rem, step), want), (rem, step)


def test_the_reference_simd_protokernels_are_close_to_its_generic_one_not_equal():
    """what "bit for bit" is claimed against: the generic protokernel.  The SSE3 / AVX protokernels of the same kernels (what a volk_gnsssdr build would
    dispatch on x86) interleave four partial sums and advance the phase in another schedule; on sums far from saturation they land within a few units."""
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_mcorr16_run") or not R.ref_simd_supported():
        pytest.skip("oracle/_ref with the SIMD flavour not available here")
    rng = np.random.default_rng(7)
    code = np.stack([oracle.ca_code(3), np.zeros(1023, np.float32)], -1).astype(np.int16)
    worst, differ = 0, 0
    for _ in range(30):
        x = rng.integers(-30, 31, size=(8000, 2)).astype(np.int16)
        args = (code, np.array([-0.5, 0, 0.5], np.float32), x, float(rng.uniform(0, 6)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(0, 1)), 0.2557)
        a, b = oracle.ref_mcorr16(*args), oracle.ref_mcorr16(*args, simd=True)
        worst = max(worst, int(np.max(np.abs(a.astype(int) - b.astype(int)))))
        differ += int(not np.array_equal(a, b))
    assert differ > 0 and worst <= 8, (differ, worst)


# ---- the algebra of the kernel's parallel saturating sums (csrc/multicorrelator_16sc.hip: Map16, map_push, map_then, map_canon) ---------------------------

def _push(g, p):
    a, lo, hi = g
    return (a + p, min(max(lo + p, -32768), 32767), min(max(hi + p, -32768), 32767))


def _then(g1, g2):
    a = min(max(g1[0] + g2[0], -65535), 65535)
    lo = min(max(g1[1] + g2[0], g2[1]), g2[2])
    hi = min(max(g1[2] + g2[0], g2[1]), g2[2])
    return (a, lo, hi)


def _canon(g):
    return (min(max(g[0], -65535), 65535), g[1], g[2])


def _at(g, x):
    return min(max(x + g[0], g[1]), g[2])


def _sequential(terms, x=0):
    for p in terms:
        x = min(max(x + int(p), -32768), 32767)
    return x


@pytest.mark.parametrize("amp", [3, 400, 20000, 32767])
def test_composed_clamp_maps_equal_the_sequential_saturating_sum(amp):
    rng = np.random.default_rng(amp)
    for trial in range(60):
        n = int(rng.integers(1, 3000))
        bias = int(rng.integers(-amp, amp + 1)) // 2
        terms = np.clip(rng.integers(-amp, amp + 1, n) + bias * (rng.random(n) < 0.5), -32768, 32767)
        want = _sequential(terms)
        # lanes own contiguous runs of unequal lengths (some empty), fold them as a tree in order -- the kernel's shape
        cuts = np.sort(rng.integers(0, n + 1, 63))
        runs = np.split(terms, cuts)
        maps = []
        for r in runs:
            g = (0, -32768, 32767)
            for p in r:
                g = _push(g, int(p))
            maps.append(_canon(g))
        while len(maps) > 1:
            maps = [_then(maps[i], maps[i + 1]) if i + 1 < len(maps) else maps[i] for i in range(0, len(maps), 2)]
        assert _at(maps[0], 0) == want, (amp, trial)
        # and from any starting value, not just 0
        for x0 in (-32768, -1, 12345, 32767):
            assert _at(maps[0], x0) == _sequential(terms, x0)
