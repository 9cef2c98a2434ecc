"""CPU model of the closed-loop kernel's seed tables (csrc/mcorr_device.h: seed_table_fill / run_segment_packed, csrc/tracking_loop.hip: GSH_TRK_SEED_TABLES).

A lane's carrier phasor at the first sample of its slice, exp(-j (rem + (n_first + 2 tid) step)), used to be ONE double-precision evaluation per lane and period
(rounded once to float32).  The kernel now takes it as the product A[odd] * W[tid / 64] * B[tid % 64] of three float32 table entries -- each evaluated once per period
by an otherwise idle wave -- with two float32 complex products.  This model restates both forms in numpy and bounds what the factorisation costs: the phase error of a
seed stays below 1e-6 rad (three roundings instead of one), far inside the 1e-5 relative bar the correlator outputs are held to."""
import numpy as np


def _expmj32(ph):
    """exp(-j ph) evaluated in double, rounded once to complex64 (what expmj leaves: <= 2e-7 rad)."""
    return np.exp(-1j * np.asarray(ph, dtype=np.float64)).astype(np.complex64)


def _cmul32(a, b):
    """complex product in float32 without fused operations (the library is built with -ffp-contract=off)."""
    ar, ai, br, bi = (np.float32(v) for v in (a.real, a.imag, b.real, b.imag))
    re = np.float32(np.float32(ar * br) - np.float32(ai * bi))
    im = np.float32(np.float32(ar * bi) + np.float32(ai * br))
    return re + 1j * im


def test_factored_seed_stays_within_a_microradian_of_the_direct_one():
    rng = np.random.default_rng(0x5EED7AB)
    worst_phase, worst_mag = 0.0, 0.0
    for _ in range(40):
        step = np.float32(2.0 * np.pi * rng.uniform(-6000.0, 6000.0) / 25e6)   # carrier phase step per sample at 25 Msps, |Doppler| <= 6 kHz
        rem = np.float32(rng.uniform(-2.0 * np.pi, 2.0 * np.pi))
        sd = np.float64(step)
        for odd in (0, 1):
            a = _expmj32(np.float64(rem) - odd * sd)
            tid = np.arange(1024)
            w = _expmj32(128.0 * (tid >> 6) * sd)
            b = _expmj32(2.0 * (tid & 63) * sd)
            got = _cmul32(_cmul32(np.full(1024, a), w), b)
            want64 = np.exp(-1j * (np.float64(rem) + (2.0 * tid - odd) * sd))
            err = np.angle(got.astype(np.complex128) * np.conj(want64))
            worst_phase = max(worst_phase, float(np.max(np.abs(err))))
            worst_mag = max(worst_mag, float(np.max(np.abs(np.abs(got.astype(np.complex128)) - 1.0))))
    assert worst_phase < 1.0e-6, worst_phase
    assert worst_mag < 1.0e-6, worst_mag


def test_the_tables_cover_every_lane_and_both_window_parities():
    """index arithmetic of the tables: lane factors 0..63, wave factors 64..79, the two parities 80..81, the three wave-uniform rotations 82..84 (mcorr_device.h)."""
    SEED_B, SEED_W, SEED_A, SEED_INC, SEED_ENTRIES = 0, 64, 80, 82, 88
    used = set()
    for lane in range(64):                       # the first table wave
        used.add(SEED_B + lane)
    for lane in range(21):                       # the second: 16 wave factors, two parities, three rotations
        used.add(SEED_W + lane if lane < 16 else (SEED_A + lane - 16 if lane < 18 else SEED_INC + lane - 18))
    assert used == set(range(85)) and max(used) < SEED_ENTRIES
    for tid in range(1024):                      # what a lane reads
        for odd in (0, 1):
            assert {SEED_B + (tid & 63), SEED_W + (tid >> 6), SEED_A + odd} <= used
