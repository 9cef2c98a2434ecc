"""GPS L1 C/A spreading codes for synthetic inputs (bench.py, examples): the Gold codes of IS-GPS-200 (G1 = 1 + x^3 + x^10,
G2 = 1 + x^2 + x^3 + x^6 + x^8 + x^9 + x^10, both registers all ones, PRN i = G1 xor G2 delayed by the PRN's chip delay).
Independent of oracle/ (the bench may only use the oracle as checker and CPU baseline); tests/test_codes.py holds these equal to the
reference's generator (gps_sdr_signal_replica.cc:24-97, 135-173) for every PRN.  In a receiver the adapters call the reference's own
generators; nothing on the GPU path depends on this module."""
from __future__ import annotations

import functools

import numpy as np

# G2 delay in chips, PRN 1..32 (IS-GPS-200 Table 3-Ia)
G2_DELAY = (5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516,
            859, 860, 861, 862)


def _lfsr(taps) -> np.ndarray:
    """1023 output bits of a 10-stage register that starts all ones; `taps` are the stages (1-based) xor-ed into the feedback."""
    reg = [1] * 10                       # reg[0] = stage 1 ... reg[9] = stage 10 (the output)
    out = np.empty(1023, np.uint8)
    for i in range(1023):
        out[i] = reg[9]
        fb = 0
        for t in taps:
            fb ^= reg[t - 1]
        reg = [fb] + reg[:9]
    return out


@functools.lru_cache(maxsize=None)
def _g1_g2():
    return _lfsr((3, 10)), _lfsr((2, 3, 6, 8, 9, 10))


def gps_l1_ca_code(prn: int) -> np.ndarray:
    """1023 chips of +-1 (float32); chip = +1 where G1 xor G2 is 1."""
    if not 1 <= prn <= 32:
        raise ValueError(f"GPS PRN {prn} outside 1..32")
    g1, g2 = _g1_g2()
    bits = g1 ^ np.roll(g2, G2_DELAY[prn - 1])
    return (2.0 * bits.astype(np.float32) - 1.0).astype(np.float32)


def gps_l1_ca_code_sampled(prn: int, fs: int) -> np.ndarray:
    """One code period sampled at fs as complex64, chip index floor(ts * i / tc) in float32 with the last sample pinned to the last chip
    (the digitisation of gps_l1_ca_code_gen_complex_sampled; like gps_l1_ca_code_gen_complex, :106-132, the chips sit in the
    imaginary part)."""
    code = gps_l1_ca_code(prn)
    n = int(float(fs) / (1023000.0 / 1023.0))
    ts = np.float32(1.0) / np.float32(fs)
    tc = np.float32(1.0) / np.float32(1023000)
    idx = np.floor((ts * np.arange(n, dtype=np.float32)) / tc).astype(np.int64)
    idx[-1] = 1022
    return (1j * code[idx]).astype(np.complex64)
