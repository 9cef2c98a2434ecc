"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the frequency-translating decimating FIR filter the input-filter adapters instantiate
(src/algorithms/input_filter/adapters/freq_xlating_fir_filter.cc:115-161, fir_filter.cc).

Parity status: UNPINNED.  The arithmetic lives in GNU Radio (gr::filter::freq_xlating_fir_filter_ccf / fir_filter_ccf over VOLK dot products;
min version 3.7.3, neither vendored in /root/reference nor installed here) and the reference holds no test vector for it.  What is restated is the
documented definition -- translate the band at center_freq to 0 Hz, low-pass with real taps, keep every D-th output, zero history before the
stream -- evaluated in float64 as the arbiter."""
import numpy as np


def freq_xlating_fir(x: np.ndarray, taps: np.ndarray, decimation: int = 1, center_freq_hz: float = 0.0, sampling_freq_hz: float = 1.0) -> np.ndarray:
    """y[m] = sum_k taps[k] x[mD - k] exp(-j 2 pi fc (mD - k) / fs), x[n] = 0 for n < 0; one output for every m with mD < len(x)."""
    x = np.asarray(x).astype(np.complex128)
    n = np.arange(len(x), dtype=np.float64)
    rev = center_freq_hz / sampling_freq_hz * n
    rev -= np.rint(rev)
    xt = x * np.exp(-2j * np.pi * rev)
    full = np.convolve(xt, np.asarray(taps, np.float64))[:len(x)]      # full[n] = sum_k h[k] xt[n - k]
    return full[::decimation]
