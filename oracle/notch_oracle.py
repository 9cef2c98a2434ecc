"""TEST INFRASTRUCTURE ONLY -- CPU restatement of gnss-sdr's notch input filters, statement by statement in float32:
  NotchOracle      Notch      src/algorithms/input_filter/gnuradio_blocks/notch_cc.cc:33-140       ("notch.cc")
  NotchLiteOracle  NotchLite  src/algorithms/input_filter/gnuradio_blocks/notch_lite_cc.cc:30-150  ("lite.cc")

Both cut the stream into segments of `length` samples.  While the noise floor is being estimated (the first n_segments_est segments after a
reset, and only while the filter is not engaged) a segment is copied through and its spectral noise floor -- FFT, power spectrum in dB, mean of
the bins not more than 15 dB above the mean -- updates the running estimate.  Afterwards a segment whose energy over the estimate exceeds the
chi-squared threshold is filtered by the one-pole notch  out[n] = in[n] - z0 in[n-1] + p z0 out[n-1]:  Notch takes z0 per SAMPLE from the
phase of in[n] conj(in[n-1]); NotchLite re-estimates one z0 every n_segments_coeff filtered segments from the phase steps at the two ends of
the segment.  out[n-1] starts at 0 whenever the filter engages.

Parity status: PINNED to the reference blocks themselves (oracle/_ref/libgnsssdr_ref_filt.so, tests/test_notch_oracle_pinned.py).  What that
library does not take from the reference: the FFT behind gr::fft (oracle/ref_fft.cc, or scipy's float32 transform through its hook) and the
VOLK kernels (not vendored: restated as their documented generic loops in oracle/shim_blocks/volk/volk.h); this file calls the same
definitions.  The segment energy is summed in float64 here and sequentially in float32 there (last-bit differences of a sum of 32 terms)."""
from __future__ import annotations

import numpy as np
from scipy.stats import chi2

from oracle.pcps_oracle import cmul, fft_fwd

LOG2_TO_10 = np.float32(3.01029995663981209120)


def power_spectrum_db(X: np.ndarray) -> np.ndarray:
    """volk_32fc_s32f_power_spectrum_32f(out, X, 1.0, n): 10 log10(re^2 + im^2) through a base-2 logarithm (-127 for 0)"""
    X = np.asarray(X, np.complex64)
    p = (X.real * X.real + X.imag * X.imag).astype(np.float32)
    with np.errstate(divide="ignore"):
        l2 = np.where(p > 0, np.log2(p, dtype=np.float32), np.float32(-127.0)).astype(np.float32)
    return (LOG2_TO_10 * l2).astype(np.float32)


def spectral_noise_floor(db: np.ndarray, exclusion: float = 15.0) -> np.float32:
    """volk_32f_s32f_calc_spectral_noise_floor_32f: mean of the bins <= (mean of all bins + exclusion)"""
    f32 = np.float32
    s = f32(0.0)
    for v in db:
        s = f32(s + v)
    mean_amp = f32(f32(s / f32(len(db))) + f32(exclusion))
    s = f32(0.0)
    kept = len(db)
    for v in db:
        if v <= mean_amp:
            s = f32(s + v)
        else:
            kept -= 1
    return mean_amp if kept == 0 else f32(s / f32(kept))


class _NotchBase:
    def __init__(self, pfa: float, p_c_factor: float, length: int, n_segments_est: int, n_segments_reset: int):
        self.length = length
        self.n_deg_fred = 2 * length
        self.thres = np.float32(chi2.isf(float(np.float32(pfa)), self.n_deg_fred))                # notch.cc:54-55
        self.p = np.complex64(complex(np.float32(p_c_factor), 0.0))
        self.n_segments_est, self.n_segments_reset = n_segments_est, n_segments_reset
        self.noise_pow_est = np.float32(0.0)
        self.n_segments = 0
        self.filter_state = False
        self.last_out = np.complex64(0.0)
        self.z0 = np.complex64(0.0)
        self.modes = []   # per segment: 0 estimate + copy, 1 filtered, 2 copied

    def _estimate(self, seg: np.ndarray):                                                         # notch.cc:77-83
        f32 = np.float32
        db = power_spectrum_db(fft_fwd(seg))
        sig2db = spectral_noise_floor(db, 15.0)
        sig2lin = f32(f32(np.power(f32(10.0), f32(sig2db / f32(10.0)), dtype=np.float32)) / f32(self.n_deg_fred))
        self.noise_pow_est = f32(f32(f32(self.n_segments) * self.noise_pow_est + sig2lin) / f32(self.n_segments + 1))

    @staticmethod
    def _energy(seg: np.ndarray) -> np.float32:                                                   # real part of volk_32fc_x2_conjugate_dot_prod_32fc(in, in)
        t = (seg.real * seg.real + seg.imag * seg.imag).astype(np.float32)
        return np.float32(np.sum(t, dtype=np.float64))

    def _one_pole(self, cur: np.ndarray, prev: np.ndarray, z0) -> np.ndarray:
        """out[n] = in[n] - z0[n] in[n-1] + (p z0[n]) last_out, std::complex<float> arithmetic in the order the expression is written"""
        out = np.empty(len(cur), np.complex64)
        z = np.broadcast_to(np.asarray(z0, np.complex64), cur.shape)
        a = (cur - cmul(z, prev)).astype(np.complex64)
        b = cmul(np.full(len(cur), self.p, np.complex64), z)
        last = self.last_out
        for i in range(len(cur)):
            last = np.complex64(a[i] + cmul(np.array([b[i]], np.complex64), np.array([last], np.complex64))[0])
            out[i] = last
        self.last_out = last
        return out


class NotchOracle(_NotchBase):
    def __init__(self, pfa: float = 0.001, p_c_factor: float = 0.9, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000):
        super().__init__(pfa, p_c_factor, length, n_segments_est, n_segments_reset)

    def general_work(self, x: np.ndarray, noutput_items: int | None = None):
        """x[0] is the sample in front of the first one processed (notch.cc:71 `in++`); returns (outputs, consumed)."""
        x = np.asarray(x, np.complex64)
        nout = len(x) if noutput_items is None else noutput_items
        L = self.length
        out = []
        idx = 0
        while idx + L < nout:                                                                     # :72
            cur, prev = x[idx + 1:idx + 1 + L], x[idx:idx + L]
            if self.n_segments < self.n_segments_est and not self.filter_state:                   # :74
                self._estimate(cur)
                out.append(cur.copy())
                self.modes.append(0)
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = self._energy(cur) / self.noise_pow_est
                if ratio > self.thres:                                                            # :88
                    if not self.filter_state:
                        self.filter_state = True
                        self.last_out = np.complex64(0.0)
                    c = cmul(cur, np.conj(prev))                                                  # :95
                    ang = np.arctan2(c.imag, c.real).astype(np.float32)                           # :96
                    z0 = (np.cos(ang, dtype=np.float32) + 1j * np.sin(ang, dtype=np.float32)).astype(np.complex64)   # :99 std::exp(j angle)
                    out.append(self._one_pole(cur, prev, z0))
                    self.z0 = z0[-1]
                    self.modes.append(1)
                else:
                    if self.n_segments > self.n_segments_reset:                                   # :106
                        self.n_segments = 0
                    self.filter_state = False
                    out.append(cur.copy())
                    self.modes.append(2)
            idx += L
            self.n_segments += 1
        return (np.concatenate(out) if out else np.zeros(0, np.complex64)), idx


class NotchLiteOracle(_NotchBase):
    def __init__(self, p_c_factor: float = 0.9, pfa: float = 0.001, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000,
                 n_segments_coeff: int = 1):
        super().__init__(pfa, p_c_factor, length, n_segments_est, n_segments_reset)
        self.n_segments_coeff_reset = n_segments_coeff
        self.n_segments_coeff = 0

    def general_work(self, x: np.ndarray, noutput_items: int | None = None):
        """x[0] is the history item (set_history(2), lite.cc:59); returns (outputs, consumed)."""
        x = np.asarray(x, np.complex64)
        nout = len(x) if noutput_items is None else noutput_items
        L = self.length
        f32 = np.float32
        out = []
        idx = 0
        while idx + L < nout:                                                                     # :81
            cur, prev = x[idx + 1:idx + 1 + L], x[idx:idx + L]
            if self.n_segments < self.n_segments_est and not self.filter_state:
                self._estimate(cur)
                out.append(cur.copy())
                self.modes.append(0)
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = self._energy(cur) / self.noise_pow_est
                if ratio > self.thres:
                    if not self.filter_state:
                        self.filter_state = True
                        self.last_out = np.complex64(0.0)
                        self.n_segments_coeff = 0
                    if self.n_segments_coeff == 0:                                                # :104-112
                        c1 = cmul(cur[1:2], np.conj(cur[0:1]))[0]
                        c2 = cmul(cur[L - 1:L], np.conj(cur[L - 2:L - 1]))[0]
                        a1 = f32(np.arctan2(f32(c1.imag), f32(c1.real)))
                        a2 = f32(np.arctan2(f32(c2.imag), f32(c2.real)))
                        ang = f32(f32(a1 + a2) / f32(2.0))
                        self.z0 = np.complex64(complex(np.cos(ang, dtype=np.float32), np.sin(ang, dtype=np.float32)))
                    out.append(self._one_pole(cur, prev, self.z0))
                    self.n_segments_coeff = (self.n_segments_coeff + 1) % self.n_segments_coeff_reset   # :118-119
                    self.modes.append(1)
                else:
                    if self.n_segments > self.n_segments_reset:
                        self.n_segments = 0
                    self.filter_state = False
                    out.append(cur.copy())
                    self.modes.append(2)
            idx += L
            self.n_segments += 1
        return (np.concatenate(out) if out else np.zeros(0, np.complex64)), idx
