"""TEST INFRASTRUCTURE ONLY -- CPU restatement of gnss-sdr's pulse blanking input filter
(src/algorithms/input_filter/gnuradio_blocks/pulse_blanking_cc.cc:33-106), statement by statement.

Parity status: PINNED since round 2 -- the block itself (compiled in place into oracle/_ref/libgnsssdr_ref_filt.so against the GNU Radio mock, its two
VOLK calls as their generic loops) gives the same outputs, bit for bit, the same consumed counts for every partition of the stream into scheduler calls and the
same noise estimate to float round-off (tests/test_notch_oracle_pinned.py::test_pulse_blanking_block).  The per-sample |x|^2 is the float expression VOLK's
generic kernel forms; the segment sum is taken in float64 and rounded once (the reference's accumulator adds in float32 in an ISA-dependent lane order).
thres_ uses scipy's chi-squared survival inverse in place of boost::math::quantile(complement).
"""
from __future__ import annotations

import numpy as np
from scipy.stats import chi2


class PulseBlankingOracle:
    def __init__(self, pfa: float = 0.04, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000):
        self.length = length
        self.n_segments_est, self.n_segments_reset = n_segments_est, n_segments_reset
        self.n_deg_fred = 2 * length                                                              # :43
        self.thres = np.float32(chi2.isf(float(np.float32(pfa)), self.n_deg_fred))                # :48-49
        self.noise_power_estimation = np.float32(0.0)                                             # :38
        self.n_segments = 0
        self.last_filtered = False

    def general_work(self, x: np.ndarray):
        """One call over the items x; returns (output samples, items consumed).  :57-106"""
        x = np.asarray(x, np.complex64)
        n = len(x)
        L = self.length
        mag = (x.real * x.real + x.imag * x.imag).astype(np.float32)                              # :63
        out = []
        idx = 0
        while idx + L < n:                                                                        # :66
            seg_e = np.float32(np.sum(mag[idx:idx + L], dtype=np.float64))                        # :68
            if self.n_segments < self.n_segments_est and not self.last_filtered:                  # :69
                self.noise_power_estimation = np.float32(
                    (np.float32(self.n_segments) * self.noise_power_estimation + seg_e / np.float32(self.n_deg_fred)) / np.float32(self.n_segments + 1))  # :71
                out.append(x[idx:idx + L])
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = seg_e / self.noise_power_estimation
                if ratio > self.thres:                                                            # :76
                    out.append(np.zeros(L, np.complex64))
                    self.last_filtered = True
                else:
                    out.append(x[idx:idx + L])
                    self.last_filtered = False
                    if self.n_segments > self.n_segments_reset:                                   # :85
                        self.n_segments = 0
            idx += L
            self.n_segments += 1                                                                  # :94
        return (np.concatenate(out) if out else np.zeros(0, np.complex64)), idx
