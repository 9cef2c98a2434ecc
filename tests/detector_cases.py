"""Shared signal builders for the Tong / Galileo 8 ms detector tests (CPU oracle tests and GPU parity tests)."""
import numpy as np

import oracle
from helpers import add_code_signal, cn0_to_amplitude, golden_e1_l5_codes, synth_gps_l1_stream

FS = 4000000


def tong_case(n_blocks: int = 12, signal: bool = True, seed: int = 2013):
    """gps_l1_ca_pcps_tong_acquisition_gsoc2013_test.cc:199-258: PRN 10, 750 Hz, 600 chips, 44 dB-Hz, 4 Msps, 1 ms blocks,
    doppler_max 10000, step 250, threshold 0.00108, tong_init_val 1, tong_max_val 8 (tong_max_dwells defaults to max_val + 1)."""
    n = 4000
    x = synth_gps_l1_stream(n_blocks * n, FS, [10] if signal else [], [750.0] if signal else [], [1023.0 - 600.0] if signal else [],
                            cn0_dbhz=44.0, seed_noise=seed)
    kw = dict(fs_in=FS, fft_size=n, doppler_max=10000, doppler_step=250, samples_per_code=4000.0, threshold=0.00108,
              tong_init_val=1, tong_max_val=8, tong_max_dwells=9)
    return x, kw, oracle.ca_code_complex_sampled(10, FS)


def e1b_local_code_8ms(prn: int) -> np.ndarray:
    """two 4 ms periods of the E1B sinBOC(1,1) replica at 4 Msps (what the adapter hands to set_local_code with
    coherent_integration_time_ms = 8: base_pcps_acquisition_custom.cc:79-81 num_codes = 2)."""
    e1b = golden_e1_l5_codes()["e1b"][prn - 1]
    idx = np.floor(np.arange(16000) * (8184.0 / 16000.0)).astype(np.int64)
    one = e1b[idx].astype(np.complex64)
    return np.concatenate([one, one])


def e1_8ms_case(flip: bool, prn: int = 11, cn0: float = 44.0, seed: int = 8, signal: bool = True):
    """galileo_e1_pcps_8ms_ambiguous_acquisition_gsoc2013_test.cc:203-259: 8 ms at 4 Msps, 750 Hz, delay 600 chips,
    doppler_max 10000, step 250, max_dwells 1; threshold from pfa = 0.1 as in the test's second configuration (:341) through
    ThresholdComputeDoppler (base_pcps_acquisition_custom.cc:89-112).  `flip`: the data symbol changes sign between the two periods."""
    n = 32000
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(2 * n) + 1j * rng.standard_normal(2 * n)).astype(np.complex64)
    if signal:
        e1b = golden_e1_l5_codes()["e1b"][prn - 1]
        delay_samples = 600.0 * (4000000.0 / 1.023e6)  # the block starts `delay` into the code
        amp = cn0_to_amplitude(cn0, FS)
        nn = np.arange(2 * n, dtype=np.float64)
        idx = np.floor((nn - delay_samples) * (8184.0 / 16000.0)).astype(np.int64) % 8184
        period = np.floor((nn - delay_samples) / 16000.0).astype(np.int64)
        sym = np.where((period % 2 == 1) & flip, -1.0, 1.0)
        x += (amp * sym * e1b[idx] * np.exp(2j * np.pi * 750.0 / FS * nn)).astype(np.complex64)
    val = (1.0 - 0.1) ** (1.0 / (n * 81))
    threshold = float(np.float32(-np.log1p(-val) / n))
    kw = dict(fs_in=FS, fft_size=n, doppler_max=10000, doppler_step=250, samples_per_code=16000.0, threshold=threshold, max_dwells=1)
    return x, kw, e1b_local_code_8ms(prn), delay_samples if signal else 0.0


def e1_local_code_4ms(prn: int, component: str) -> np.ndarray:
    """one 4 ms period of the E1B ("e1b") or E1C ("e1c") sinBOC(1,1) replica at 4 Msps (what
    galileo_e1_pcps_cccwsr_ambiguous_acquisition.cc:57-69 hands to set_local_code: data = 1B, pilot = 1C)."""
    c = golden_e1_l5_codes()[component][prn - 1]
    idx = np.floor(np.arange(16000) * (8184.0 / 16000.0)).astype(np.int64)
    return c[idx].astype(np.complex64)


def cccwsr_case(mode: str = "inphase", data_sign: float = 1.0, pilot_sign: float = -1.0, prn: int = 10, cn0: float = 44.0,
                seed: int = 13, signal: bool = True, n_blocks: int = 2):
    """galileo_e1_pcps_cccwsr_ambiguous_acquisition_gsoc2013_test.cc:265-350 (config_2): Galileo PRN 10, 4 ms at 4 Msps, 750 Hz, delay 600 chips,
    44 dB-Hz, doppler_max 10000, step 250, max_dwells 1, threshold 0.00215.
    mode "inphase": data_sign*E1B + pilot_sign*E1C on one carrier phase (the E1 composite is E1B - E1C) -- both combining branches
    then have equal expected peaks.  mode "quadrature": the pilot is rotated by +90 degrees, so exactly one branch combines coherently
    (data + j*pilot sees data_sign - pilot_sign, data - j*pilot sees data_sign + pilot_sign)."""
    n = 16000
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n_blocks * n) + 1j * rng.standard_normal(n_blocks * n)).astype(np.complex64)
    delay_samples = 600.0 * (4000000.0 / 1.023e6)
    if signal:
        g = golden_e1_l5_codes()
        amp = cn0_to_amplitude(cn0, FS)
        nn = np.arange(n_blocks * n, dtype=np.float64)
        idx = np.floor((nn - delay_samples) * (8184.0 / 16000.0)).astype(np.int64) % 8184
        rot = 1.0 if mode == "inphase" else 1j
        comp = data_sign * g["e1b"][prn - 1][idx] + rot * pilot_sign * g["e1c"][prn - 1][idx]
        x += (amp * comp * np.exp(2j * np.pi * 750.0 / FS * nn)).astype(np.complex64)
    kw = dict(fs_in=FS, fft_size=n, doppler_max=10000, doppler_step=250, samples_per_code=16000.0, threshold=0.00215, max_dwells=1)
    return x, kw, e1_local_code_4ms(prn, "e1b"), e1_local_code_4ms(prn, "e1c"), delay_samples if signal else 0.0


def quicksync_case(fs: int = 8000000, folding_factor: int = 4, signal: bool = True, seed: int = 2014):
    """gps_l1_ca_pcps_quicksync_acquisition_gsoc2014_test.cc:222-279: PRN 10, 750 Hz, 600 chips, 44 dB-Hz, fs 8 Msps, 4 ms, the
    adapter's default folding factor ceil(sqrt(log2(8000))) = 4 (gps_l1_ca_pcps_quicksync_acquisition.cc:39), doppler_max 10000,
    step 250."""
    spc = fs // 1000
    n = spc * folding_factor
    x = synth_gps_l1_stream(n, fs, [10] if signal else [], [750.0] if signal else [], [1023.0 - 600.0] if signal else [], cn0_dbhz=44.0,
                            seed_noise=seed)
    # the block's statistic is not normalised for the folding (noise floor ~ p^3 / fft_size), so the threshold is per configuration
    threshold = {(8000000, 4): 0.7, (4000000, 4): 1.0, (8000000, 2): 0.033}[(fs, folding_factor)]
    kw = dict(fs_in=fs, samples_per_code=spc, folding_factor=folding_factor, doppler_max=10000, doppler_step=250, threshold=threshold, max_dwells=1)
    return x, kw, oracle.ca_code_complex_sampled(10, fs)


def fine_doppler_case(doppler_hz: float = 1730.0, fs: int = 4000000, seed: int = 77, signal: bool = True):
    """GPS L1 C/A, PRN 10, 600 chips, 47 dB-Hz, 12 ms of signal at 4 Msps; doppler_max 5000, step 500, 2 dwells, peak-ratio threshold 2.5."""
    n = fs // 1000
    x = synth_gps_l1_stream(12 * n, fs, [10] if signal else [], [doppler_hz] if signal else [], [1023.0 - 600.0] if signal else [], cn0_dbhz=47.0,
                            seed_noise=seed)
    kw = dict(fs_in=fs, samples_per_ms=float(n), doppler_max=5000, doppler_step=500, threshold=2.5, max_dwells=2)
    return x, kw, oracle.ca_code_complex_sampled(10, fs)


def e5a_local_codes(fs: int, prn: int, sampled_ms: int, zero_padding: int = 0):
    """What galileo_e5a_noncoherent_iq_acquisition_caf.cc:94-141 hands to the block: the data (5I) and pilot (5Q) primary codes sampled at fs
    (galileo_e5_a_code_gen_complex_sampled), one code period per millisecond, `sampled_ms` periods (or one period + zeros with Zero_padding).
    The 10 230-chip L5 I / Q golden codes stand in for the E5a-I / E5a-Q primary codes (same length and chip rate; the block never looks at the
    chips, and the adapter test generates the real ones with the reference's generator)."""
    g = golden_e1_l5_codes()
    spc = fs // 1000
    idx = np.floor(np.arange(spc) * (10230000.0 / fs)).astype(np.int64) % 10230
    one_i, one_q = g["l5i"][prn - 1][idx].astype(np.complex64), g["l5q"][prn - 1][idx].astype(np.complex64)
    n = spc * sampled_ms
    ci, cq = np.zeros(n, np.complex64), np.zeros(n, np.complex64)
    reps = 1 if zero_padding > 0 else sampled_ms
    for k in range(reps):
        ci[k * spc:(k + 1) * spc] = one_i
        cq[k * spc:(k + 1) * spc] = one_q
    return ci, cq


def e5a_case(fs: int = 12000000, sampled_ms: int = 3, both: bool = True, data_signs=(1, 1, 1, 1, 1, 1), pilot_signs=(1, 1, 1, 1, 1, 1), doppler: float = 250.0,
             delay_chips: float = 1000.0, cn0: float = 46.0, prn: int = 11, seed: int = 5, signal: bool = True, doppler_max: int = 5000,
             doppler_step: int = 250, caf_window_hz: int = 0, zero_padding: int = 0, max_dwells: int = 1, n_blocks: int = 2):
    """galileo_e5a_pcps_acquisition_gsoc2014_gensource_test.cc config_1 (:205-282: 32 Msps, 1 ms, 2800 Hz) and config_2 / config_3 (:284-430: 12 Msps,
    3 ms, delay 1000 chips, 250 Hz, doppler_max 5000 / step 250 -- 10 000 / 250 in config_1): data on I, pilot on Q (E5a = data + j pilot), each with its
    own secondary-code / symbol sign per millisecond (`data_signs`, `pilot_signs`), in unit-variance complex noise."""
    spc = fs // 1000
    n = spc * sampled_ms
    rng = np.random.default_rng(seed)
    total = n_blocks * n
    x = (rng.standard_normal(total) + 1j * rng.standard_normal(total)).astype(np.complex64)
    g = golden_e1_l5_codes()
    if signal:
        amp = cn0_to_amplitude(cn0, fs) / np.sqrt(2.0 if both else 1.0)
        nn = np.arange(total, dtype=np.float64)
        delay_samples = delay_chips * fs / 10230000.0
        chip = np.floor((nn - delay_samples) * (10230000.0 / fs)).astype(np.int64) % 10230
        ms = np.floor((nn - delay_samples) / spc).astype(np.int64)
        ds = np.asarray(data_signs, np.float64)[ms % len(data_signs)]
        ps = np.asarray(pilot_signs, np.float64)[ms % len(pilot_signs)]
        s = ds * g["l5i"][prn - 1][chip]
        if both:
            s = s + 1j * ps * g["l5q"][prn - 1][chip]
        x += (amp * s * np.exp(2j * np.pi * doppler / fs * nn)).astype(np.complex64)
    nb = 0
    d = -doppler_max
    while d <= doppler_max:
        nb += 1
        d += doppler_step
    val = (1.0 - 0.01) ** (1.0 / (n * nb))                                 # ThresholdComputeDoppler, pfa 0.01 (base_pcps_acquisition_custom.cc:89-112)
    threshold = float(np.float32(-np.log1p(-val) / n))
    kw = dict(fs_in=fs, fft_size=n, doppler_max=doppler_max, doppler_step=doppler_step, samples_per_code=spc, threshold=threshold, max_dwells=max_dwells,
              sampled_ms=sampled_ms, both_signal_components=both, caf_window_hz=caf_window_hz, zero_padding=zero_padding)
    ci, cq = e5a_local_codes(fs, prn, sampled_ms, zero_padding)
    return x, kw, ci, cq
