"""Shared scenario builders for the symbol-synchronisation tests (CPU oracle and GPU): signals with navigation data / secondary codes.

Constants come from the ICDs, not from the reference tree: the GPS LNAV telemetry word preamble is 10001011 (IS-GPS-200, 20.3.3.1), one
bit = 20 C/A code periods; the Galileo E1-C secondary code CS25_1 is 0011100000001010110110010 (Galileo OS SIS ICD, table 19).
tests/test_symbol_sync.py checks them against the reference's headers when those are present."""
import numpy as np

import oracle
from helpers import GPS_CA_CHIP_RATE, GPS_L1_FREQ_HZ, TWO_PI, cn0_to_amplitude

GPS_PREAMBLE_BITS = "10001011"
GPS_CA_PREAMBLE_SYMBOLS = "".join(b * 20 for b in GPS_PREAMBLE_BITS)          # 160 symbols, what d_secondary_code_string holds for "1C"
GALILEO_E1_C_SECONDARY_CODE = "0011100000001010110110010"


def gps_l1_with_nav_bits(n_periods, fs, prn, doppler_hz, bits, cn0_dbhz=47.0, seed=11, first_bit_period=0):
    """C/A signal whose code starts at sample 0; navigation bit k (characters '0'/'1', '1' -> +1 as the reference's telemetry decoder
    reads a positive prompt) spans code periods [first_bit_period + 20k, first_bit_period + 20k + 20)."""
    n = int(round(fs * 1e-3))
    total = (n_periods + 3) * n
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(total) + 1j * rng.standard_normal(total)
    t = np.arange(total, dtype=np.float64)
    f_code = GPS_CA_CHIP_RATE * (1.0 + doppler_hz / GPS_L1_FREQ_HZ)
    chips = t * (f_code / fs)
    code = oracle.ca_code(prn).astype(np.float64)
    period = np.floor(chips / 1023.0).astype(np.int64)
    k = (period - first_bit_period) // 20
    sym = np.array([1.0 if b == "1" else -1.0 for b in bits])
    d = np.where((k >= 0) & (k < len(bits)), sym[np.clip(k, 0, len(bits) - 1)], 1.0)
    x += cn0_to_amplitude(cn0_dbhz, fs) * d * code[np.floor(chips).astype(np.int64) % 1023] * np.exp(1j * TWO_PI * doppler_hz / fs * t)
    return x.astype(np.complex64), n


def galileo_e1_with_secondary(n_periods, fs, e1b, e1c, doppler_hz, data_bits, cn0_dbhz=47.0, seed=13):
    """E1 OS signal (sinBOC(1,1) replicas at 2 samples per chip as the tracking block uses them): (e1b * data - e1c * secondary) / sqrt(2),
    4 ms code period, code starts at sample 0; the pilot's secondary code CS25 runs from period 0."""
    n = int(round(fs * 4e-3))
    total = (n_periods + 3) * n
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(total) + 1j * rng.standard_normal(total)
    t = np.arange(total, dtype=np.float64)
    rate = 1.023e6 * (1.0 + doppler_hz / GPS_L1_FREQ_HZ) / fs * 2.0       # code samples (half chips) per input sample
    pos = t * rate
    idx = np.floor(pos).astype(np.int64) % 8184
    period = np.floor(pos / 8184.0).astype(np.int64)
    sec = np.array([1.0 if ch == "0" else -1.0 for ch in GALILEO_E1_C_SECONDARY_CODE])[period % 25]
    dat = np.array([1.0 if b == "1" else -1.0 for b in data_bits])[np.clip(period, 0, len(data_bits) - 1)]
    sig = (e1b.astype(np.float64)[idx] * dat - e1c.astype(np.float64)[idx] * sec) / np.sqrt(2.0)
    x += cn0_to_amplitude(cn0_dbhz, fs) * sig * np.exp(1j * TWO_PI * doppler_hz / fs * t)
    return x.astype(np.complex64), n
