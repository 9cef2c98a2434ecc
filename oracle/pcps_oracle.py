"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the PCPS acquisition arithmetic.

Follows pcps_acquisition (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc, cited per function as
acq.cc:line) in float32 / complex64, in the reference's order of operations.

Parity status: PINNED (round 2).  The reference's own pcps_acquisition.cc -- and pcps_tong_acquisition_cc.cc,
galileo_pcps_8ms_acquisition_cc.cc, pcps_cccwsr_acquisition_cc.cc, pcps_quicksync_acquisition_cc.cc, pcps_acquisition_fine_doppler_cc.cc --
are compiled from /root/reference into oracle/_ref/libgnsssdr_ref_acq.so (oracle/Makefile, oracle/ref_acq_api.cc) against stand-ins
for GNU Radio / VOLK / FFTW and driven through general_work; tests/test_pcps_oracle_pinned.py requires this restatement to equal
those blocks VALUE FOR VALUE (wipe-off tables, code spectra, every grid cell, indices, power, statistic, threshold, Gnss_Synchro,
state, messages) when both use the same transform (scipy's single-precision pocketfft plugged into the blocks' gr::fft objects),
and index for index when the blocks run on an exact-definition float64 DFT instead.
What stays outside the reference tree is the transform itself: gr::fft::fft_complex_fwd/rev is FFTW3f (gnss_sdr_fft.h:26-61), not
vendored, not installed; FFTW's float32 rounding is not reproduced by anybody here.  Peak INDICES -- what north_star requires
bit-exact -- do not depend on it (asserted with two different transforms).  The in-tree kernels go through oracle_sincos /
oracle_index_max (== the volk_gnsssdr `_generic` protokernels, bit-exact, tests/test_oracle_golden.py); VOLK's element-wise kernels
are restated as their `_generic` definitions (cmul below).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import scipy.fft

from . import lib

TWO_PI = np.float32(6.283185307179586)


def cmul(a: np.ndarray, b) -> np.ndarray:
    """Element-wise complex product as VOLK's generic kernels form it (volk_32fc_x2_multiply_32fc, acq.cc:531,538): four float32
    products, one subtraction, one addition, each rounded once.  (numpy's own complex64 multiply may contract to FMA.)  complex128
    operands (the `precise` arbiter) go through numpy's product."""
    a = np.asarray(a)
    if a.dtype != np.complex64:
        return a * b
    b = np.asarray(b, np.complex64)
    ar, ai, br, bi = a.real, a.imag, b.real, b.imag
    out = np.empty(np.broadcast(a, b).shape, np.complex64)
    out.real = ar * br - ai * bi
    out.imag = ar * bi + ai * br
    return out


def fft_fwd(a: np.ndarray) -> np.ndarray:
    """gr::fft::fft_complex_fwd::execute (FFTW forward c2c, e^{-j}, unnormalised)."""
    return scipy.fft.fft(a)


def fft_rev(a: np.ndarray) -> np.ndarray:
    """gr::fft::fft_complex_rev::execute (FFTW backward c2c, e^{+j}, UNNORMALISED: N times the inverse DFT)."""
    return scipy.fft.ifft(a, norm="forward")


class PcpsOracle:
    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_chip: int,
                 samples_per_code: float, num_doppler_bins: int = 0, consumed_samples: int | None = None,
                 doppler_center: int = 0, doppler_bias: int = 0, bit_transition_flag: bool = False, use_cfar: bool = True,
                 precise: bool = False):
        self.fs_in = fs_in
        self.fft_size = fft_size
        self.consumed = fft_size if consumed_samples is None else consumed_samples
        self.effective = fft_size // 2 if bit_transition_flag else fft_size                      # acq.cc:112
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.doppler_center, self.doppler_bias = doppler_center, doppler_bias
        self.n_bins = num_doppler_bins or int(math.ceil(2.0 * doppler_max / doppler_step))        # acq.cc:113
        self.spc = samples_per_chip
        self.samples_per_code = np.float32(samples_per_code)
        self.bit_transition = bit_transition_flag
        self.use_cfar = use_cfar
        self.precise = precise  # float64 evaluation (arbiter for near-ties), not the reference's arithmetic
        self.grid = None
        self.fft_codes = None
        self._wipeoffs()

    # acq.cc:275-291: D tables exp(-j 2 pi f n / fs), phase accumulated in float32 by the volk kernel
    def _wipeoffs(self):
        self.wipe = np.empty((self.n_bins, self.fft_size), np.complex128 if self.precise else np.complex64)
        self.freqs = []
        for d in range(self.n_bins):
            doppler = -int(self.doppler_max) + self.doppler_center + self.doppler_step * d       # acq.cc:288
            self.freqs.append(doppler)
            freq = np.float32(self.doppler_bias + doppler)                                        # acq.cc:289 cast
            if self.precise:
                n = np.arange(self.fft_size, dtype=np.float64)
                self.wipe[d] = np.exp(-2j * np.pi * float(self.doppler_bias + doppler) * n / self.fs_in)
                continue
            phase_step = TWO_PI * freq / np.float32(self.fs_in)                                   # acq.cc:278
            out = np.empty(2 * self.fft_size, np.float32)
            ph = C.c_float(0.0)
            lib().oracle_sincos(out, float(-phase_step), C.byref(ph), self.fft_size)              # acq.cc:280
            self.wipe[d] = out.view(np.complex64)

    def set_doppler_center(self, center: int):                                                    # acq.cc:737-746
        if center != self.doppler_center:
            self.doppler_center = center
            self._wipeoffs()

    # acq.cc:218-251
    def set_local_code(self, code: np.ndarray):
        dt = np.complex128 if self.precise else np.complex64
        buf = np.zeros(self.fft_size, dt)
        if self.bit_transition:
            off = self.fft_size // 2
            buf[off:] = code[:off]
        elif self.consumed == self.fft_size:
            buf[:] = code[:self.consumed]
        else:
            buf[self.consumed:] = code[:self.consumed]                                            # acq.cc:245-246
        self.fft_codes = np.conj(fft_fwd(buf)).astype(dt)                                   # acq.cc:249-250

    # acq.cc:522-560 with the zero padding of :657-664
    def doppler_grid(self, x: np.ndarray, dwell_count: int = 1):
        dt = np.complex128 if self.precise else np.complex64
        sig = np.zeros(self.fft_size, dt)
        sig[:self.consumed] = x[:self.consumed]
        if self.grid is None or dwell_count == 1:
            self.grid = np.zeros((self.n_bins, self.effective), np.float64 if self.precise else np.float32)
        off = self.effective if self.bit_transition else 0
        for d in range(self.n_bins):
            a = cmul(sig, self.wipe[d])                                                           # :531
            A = fft_fwd(a)                                                                        # :535
            B = cmul(A, self.fft_codes)                                                           # :538
            y = fft_rev(B)                                                                        # :541 unnormalised
            y = y.astype(dt)[off:off + self.effective]
            mag = (y.real * y.real + y.imag * y.imag)                                             # :547
            if dwell_count == 1:
                self.grid[d] = mag
            else:
                self.grid[d] = self.grid[d] + mag                                                 # :551-552
        return self.grid

    def _argmax(self, row: np.ndarray) -> int:
        if self.precise:
            return int(np.argmax(row))  # numpy returns the first maximum too
        t = np.zeros(1, np.uint32)
        lib().oracle_index_max(t, np.ascontiguousarray(row, np.float32), len(row))
        return int(t[0])

    # acq.cc:409-449 and :452-519
    def statistics(self, dwell_count: int = 1) -> dict:
        g = self.grid
        gmax, idx_d, idx_t = 0.0, 0, 0
        for d in range(self.n_bins):
            t = self._argmax(g[d])
            if g[d][t] > gmax:                                                                   # :420 strict
                gmax, idx_d, idx_t = g[d][t], d, t
        res = dict(index_time=idx_t, index_doppler=idx_d, peak=float(gmax),
                   doppler_hz=-int(self.doppler_max) + self.doppler_center + self.doppler_step * idx_d,
                   acq_delay_samples=float(np.fmod(np.float32(idx_t), self.samples_per_code)))   # :582
        if self.use_cfar:
            opp = (idx_d + self.n_bins // 2) % self.n_bins                                        # :429
            if self.precise:
                power = float(np.sum(g[opp]) / self.effective / 2.0 / dwell_count)
            else:
                s = np.float32(0.0)
                # std::accumulate in float32, sequential (:430)
                s = np.cumsum(g[opp].astype(np.float32), dtype=np.float32)[-1]
                power = float(np.float32(np.float64(s / np.float32(self.effective)) / 2.0 / dwell_count))
            res["input_power"] = power
            res["test_statistics"] = 0.0 if power < np.finfo(np.float32).eps else float(gmax) / power  # :438-445
            res["second_peak"] = 0.0
        else:
            e1, e2 = idx_t - self.spc, idx_t + self.spc                                          # :485-486
            if e1 < 0:
                e1 += self.effective
            elif e2 >= self.effective:
                e2 -= self.effective
            tmp = g[idx_d].copy()
            i = e1
            while True:                                                                           # :498-509 do-while
                tmp[i] = 0.0
                i += 1
                if i == self.effective:
                    i = 0
                if i == e2:
                    break
            second = tmp[self._argmax(tmp)]
            res["second_peak"] = float(second)
            res["input_power"] = 0.0
            res["test_statistics"] = float(np.float32(gmax) / np.float32(second)) if not self.precise else float(gmax / second)
        return res

    # ---- step two of make_two_steps -------------------------------------------------------------------------
    # acq.cc:294-301 (update_grid_doppler_wipeoffs_step2), :522-560 with d_step_two (same loop over the narrow tables,
    # rows 0..nbins2-1 of d_magnitude_grid), :428-437 / :475-482 (result.doppler, d_input_power NOT recomputed)
    def dwell_step2(self, x: np.ndarray, doppler_center_step_two: float, num_doppler_bins_step2: int, doppler_step2: float,
                    input_power_step_one: float = 0.0, dwell_count: int = 1) -> dict:
        nb2 = int(num_doppler_bins_step2)
        center = np.float32(doppler_center_step_two)
        half = np.float32(math.floor(nb2 / 2.0))
        step2 = np.float32(doppler_step2)
        saved = (self.wipe, self.n_bins, self.grid)
        wipe = np.empty((nb2, self.fft_size), np.complex128 if self.precise else np.complex64)
        freqs = []
        for d in range(nb2):
            doppler = (np.float32(d) - half) * step2                                              # acq.cc:298
            freq = np.float32(center + doppler)                                                   # acq.cc:299
            freqs.append(float(freq))
            if self.precise:
                n = np.arange(self.fft_size, dtype=np.float64)
                wipe[d] = np.exp(-2j * np.pi * float(freq) * n / self.fs_in)
                continue
            phase_step = TWO_PI * freq / np.float32(self.fs_in)                                   # acq.cc:278
            out = np.empty(2 * self.fft_size, np.float32)
            ph = C.c_float(0.0)
            lib().oracle_sincos(out, float(-phase_step), C.byref(ph), self.fft_size)              # acq.cc:280
            wipe[d] = out.view(np.complex64)
        try:
            self.wipe, self.n_bins = wipe, nb2
            if dwell_count == 1:
                self.grid = None
            elif self.grid is not None:
                self.grid = self.grid[:nb2]
            g = self.doppler_grid(x, dwell_count)
            gmax, idx_d, idx_t = 0.0, 0, 0
            for d in range(nb2):
                t = self._argmax(g[d])
                if g[d][t] > gmax:                                                               # :420 strict
                    gmax, idx_d, idx_t = g[d][t], d, t
            res = dict(index_time=idx_t, index_doppler=idx_d, peak=float(gmax),
                       doppler_hz=int(np.float32(center + (np.float32(idx_d) - half) * step2)),  # :436 static_cast<int32_t>
                       acq_delay_samples=float(np.fmod(np.float32(idx_t), self.samples_per_code)), freqs=freqs)
            if self.use_cfar:
                p = np.float32(input_power_step_one)
                res["input_power"] = float(p)
                res["test_statistics"] = 0.0 if p < np.finfo(np.float32).eps else float(np.float32(gmax) / p)   # :438-445
                res["second_peak"] = 0.0
            else:
                self.n_bins = nb2
                st = self.statistics(dwell_count)                                                 # same blanking / second scan
                res["second_peak"], res["test_statistics"], res["input_power"] = st["second_peak"], st["test_statistics"], 0.0
            res["grid"] = g
            return res
        finally:
            self.wipe, self.n_bins, self.grid = saved

    def dwell(self, x: np.ndarray, dwell_count: int = 1) -> dict:
        self.doppler_grid(x, dwell_count)
        return self.statistics(dwell_count)


def compute_threshold(pfa: float, effective_fft_size: int, num_doppler_bins: int, max_dwells: int) -> float:
    """acq.cc:52-56 with scipy.special.gammaincinv == boost::math::gamma_p_inv."""
    from scipy.special import gammaincinv
    num_bins = effective_fft_size * num_doppler_bins
    prob = (1.0 - float(np.float32(pfa))) ** (1.0 / float(np.float32(num_bins)))
    return float(np.float32(2.0 * gammaincinv(2.0 * max_dwells, prob)))


# =====================================================================================================================
# The other PCPS detectors.  Both reuse PcpsOracle's pinned pieces (wipe-off tables through oracle_sincos,
# arg-max through oracle_index_max); the FFT stays the unpinned scipy one, as above.
# =====================================================================================================================
def count_doppler_bins(doppler_max: int, doppler_step: int) -> int:
    """pcps_tong_acquisition_cc.cc:97-100 / galileo_pcps_8ms_acquisition_cc.cc:65-68: bins counted INCLUSIVE of +doppler_max
    (the main PCPS block uses ceil(2*max/step), acq.cc:113)."""
    n, d = 0, -doppler_max
    while d <= doppler_max:
        n += 1
        d += doppler_step
    return n


def mean_input_power(x: np.ndarray) -> np.float32:
    """tong.cc:208-210 / 8ms.cc:190-192: float |x|^2 per sample (volk_32fc_magnitude_squared_32f), summed, / (float) N.
    The reference sums in float32 in the lane order of whichever volk_32f_accumulator_s32f kernel its machine dispatches
    (unpinned); here the float terms are summed in float64 and rounded once."""
    x = np.asarray(x, np.complex64)
    terms = (x.real * x.real + x.imag * x.imag).astype(np.float32)
    return np.float32(np.float32(np.sum(terms, dtype=np.float64)) / np.float32(len(x)))


class TongOracle:
    """pcps_tong_acquisition_cc::general_work (pcps_tong_acquisition_cc.cc:147-400) for one channel: every dwell adds the
    power-normalised |y|^2 of the new block to d_grid_data; the grid maximum is compared with threshold * dwell_count and
    moves the Tong counter up or down."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 tong_init_val: int, tong_max_val: int, tong_max_dwells: int):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.p = PcpsOracle(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, num_doppler_bins=self.n_bins)
        self.fft_size = fft_size
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.tong_init_val, self.tong_max_val, self.tong_max_dwells = tong_init_val, tong_max_val, tong_max_dwells
        self.state = 0
        self.init()

    def set_local_code(self, code: np.ndarray):                                                  # tong.cc:136-144
        self.p.set_local_code(code)

    def init(self):                                                                              # tong.cc:162-184 (state 0)
        self.dwell_count = 0
        self.tong_count = self.tong_init_val
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.grid = np.zeros((self.n_bins, self.fft_size), np.float32)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # tong.cc:187-301 (state 1)
        p = self.p
        x = np.asarray(x[:self.fft_size], np.complex64)
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)                               # :195
        self.mag = np.float32(0.0)
        self.dwell_count += 1
        self.input_power = mean_input_power(x)                                                    # :208-210
        weight = np.float32(1.0) / (fnf * fnf * self.input_power)                                 # :245: N^4 P, the statistic is the captured fraction of the block's power
        self.weight = weight
        for d in range(self.n_bins):
            doppler = -int(p.doppler_max) + p.doppler_step * d                                    # :216
            a = cmul(x, p.wipe[d])                                                                # :218
            A = fft_fwd(a)
            B = cmul(A, p.fft_codes)                                                              # :227
            y = fft_rev(B)                                                                        # :231 unnormalised
            mag = (y.real * y.real + y.imag * y.imag).astype(np.float32)                          # :234
            mag = (mag * weight).astype(np.float32)                                               # :244-246
            self.grid[d] = self.grid[d] + mag                                                     # :249
            t = p._argmax(self.grid[d])                                                           # :252
            magt = self.grid[d][t]
            if self.mag < magt:                                                                   # :257 strict
                self.mag = magt
                self.result = dict(acq_delay_samples=float(t % self.samples_per_code), doppler_hz=float(doppler),
                                   doppler_step=p.doppler_step, index_time=t, index_doppler=d)
        self.test_statistics = self.mag                                                           # :277
        if self.test_statistics > self.threshold * np.float32(self.dwell_count):                  # :279
            self.tong_count += 1
            if self.tong_count == self.tong_max_val:
                self.state = 2
        else:
            self.tong_count -= 1
            if self.tong_count == 0:
                self.state = 3
        if self.dwell_count >= self.tong_max_dwells:                                              # :296
            self.state = 3
        return self.state


class Galileo8msOracle:
    """galileo_pcps_8ms_acquisition_cc::general_work (galileo_pcps_8ms_acquisition_cc.cc:134-300): the 8 ms block is
    correlated with two local codes -- A = the two 4 ms primary-code periods as generated, B = the second period
    sign-inverted (a secondary-code / data transition in the middle) -- and the larger per-bin maximum is kept."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 max_dwells: int):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.pa = PcpsOracle(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, num_doppler_bins=self.n_bins)
        self.fft_size = fft_size
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.init()

    def set_local_code(self, code: np.ndarray):                                                  # 8ms.cc:103-131
        code = np.asarray(code[:self.fft_size], np.complex64)
        self.pa.set_local_code(code)
        self.fft_code_a = self.pa.fft_codes
        b = code.copy()
        spc = self.samples_per_code
        b[spc:2 * spc] = b[spc:2 * spc] * np.complex64(-1.0)                                      # :114-123
        self.pa.set_local_code(b)
        self.fft_code_b = self.pa.fft_codes

    def init(self):                                                                              # 8ms.cc:147-160 (state 0)
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # 8ms.cc:163-287 (state 1)
        p = self.pa
        x = np.asarray(x[:self.fft_size], np.complex64)
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)
        self.mag = np.float32(0.0)
        self.well_count += 1
        self.input_power = mean_input_power(x)                                                    # :190-192
        self.rows = []
        for d in range(self.n_bins):
            doppler = -int(p.doppler_max) + p.doppler_step * d
            a = cmul(x, p.wipe[d])
            A = fft_fwd(a)
            best = []
            for codes in (self.fft_code_a, self.fft_code_b):
                B = cmul(A, codes)
                y = fft_rev(B)
                mag = (y.real * y.real + y.imag * y.imag).astype(np.float32)
                t = p._argmax(mag)
                best.append((np.float32(mag[t] / (fnf * fnf)), t))                                # :222, :237
            (ma, ta), (mb, tb) = best
            magt, t, which = (ma, ta, 0) if ma >= mb else (mb, tb, 1)                             # :240-249
            self.rows.append((float(ma), ta, float(mb), tb))
            if self.mag < magt:                                                                   # :252 strict
                self.mag = magt
                self.result = dict(acq_delay_samples=float(t % self.samples_per_code), doppler_hz=float(doppler),
                                   doppler_step=p.doppler_step, index_time=t, index_doppler=d, code=which)
        self.test_statistics = np.float32(self.mag / self.input_power)                            # :278
        if self.test_statistics > self.threshold:
            self.state = 2
        elif self.well_count == self.max_dwells:
            self.state = 3
        return self.state


class CccwsrOracle:
    """pcps_cccwsr_acquisition_cc::general_work (pcps_cccwsr_acquisition_cc.cc:137-373, cited as cccwsr.cc:line): coherent
    channel combining with sign recovery.  Per Doppler bin the wiped-off block is correlated with the data code and with
    the pilot code SEPARATELY (two inverse transforms), the two complex correlations are combined as data + j*pilot and
    data - j*pilot, and the larger of the two maxima is kept.  Restated literally -- the product uses one transform per
    branch with pre-combined local codes, so parity against this class also checks that identity.
    Reference behaviour kept as is: d_mag is cleared in state 0 only (cccwsr.cc:160), so it is a running maximum over the
    dwells of one acquisition, while d_input_power is overwritten by every dwell (cccwsr.cc:192-194).
    Note for tests: when data and pilot components share one carrier phase the two branches have EQUAL expected peaks
    (|b + j s|^2 == |b - j s|^2), so which branch wins is decided by rounding; compare what the block publishes (delay,
    Doppler, statistic, state), not the branch."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 max_dwells: int):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)                               # cccwsr.cc:79-82
        self.pa = PcpsOracle(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, num_doppler_bins=self.n_bins)
        self.fft_size = fft_size
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.init()

    def set_local_code(self, code_data: np.ndarray, code_pilot: np.ndarray):                      # cccwsr.cc:116-134
        self.pa.set_local_code(np.asarray(code_data[:self.fft_size], np.complex64))
        self.fft_code_data = self.pa.fft_codes
        self.pa.set_local_code(np.asarray(code_pilot[:self.fft_size], np.complex64))
        self.fft_code_pilot = self.pa.fft_codes

    def init(self):                                                                              # cccwsr.cc:152-164 (state 0)
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # cccwsr.cc:166-306 (state 1)
        p = self.pa
        n = self.fft_size
        x = np.asarray(x[:n], np.complex64)
        fnf = np.float32(n) * np.float32(n)                                                       # :178
        self.well_count += 1                                                                      # :182
        self.input_power = mean_input_power(x)                                                    # :192-194
        self.rows = []
        for d in range(self.n_bins):
            doppler = -int(p.doppler_max) + p.doppler_step * d                                    # :200
            A = fft_fwd(cmul(x, p.wipe[d]))                                                       # :202-207
            cd = fft_rev(cmul(A, self.fft_code_data))                                             # :212-220
            cp = fft_rev(cmul(A, self.fft_code_pilot))                                            # :225-233
            best = []
            for sgn in (np.float32(1.0), np.float32(-1.0)):                                       # :235-244
                re = (cd.real - sgn * cp.imag).astype(np.float32)
                im = (cd.imag + sgn * cp.real).astype(np.float32)
                mag = (re * re + im * im).astype(np.float32)                                      # :246, :250
                t = p._argmax(mag)                                                                # :247, :251
                best.append((np.float32(mag[t] / (fnf * fnf)), t))                                # :248, :252
            (mp_, tp), (mm, tm) = best
            magt, t, which = (mp_, tp, 0) if mp_ >= mm else (mm, tm, 1)                           # :254-263
            self.rows.append((float(mp_), tp, float(mm), tm))
            if self.mag < magt:                                                                   # :266 strict
                self.mag = magt
                self.result = dict(acq_delay_samples=float(t % self.samples_per_code), doppler_hz=float(doppler),
                                   doppler_step=p.doppler_step, index_time=t, index_doppler=d, branch=which)
        self.test_statistics = np.float32(self.mag / self.input_power)                            # :292
        if self.test_statistics > self.threshold:                                                 # :295
            self.state = 2
        elif self.well_count == self.max_dwells:                                                  # :299
            self.state = 3
        return self.state


class QuickSyncOracle:
    """pcps_quicksync_acquisition_cc::general_work (pcps_quicksync_acquisition_cc.cc:155-360), bit_transition_flag = false:
    the block of folding_factor code periods is wiped off, folded (its folding_factor^2 segments of fft_size =
    samples_per_code / folding_factor samples are added), circularly correlated with the equally folded code, and the winning
    folded delay is resolved among its folding_factor aliases by a direct correlation with the unfolded code."""

    def __init__(self, fs_in: int, samples_per_code: int, folding_factor: int, doppler_max: int, doppler_step: int, threshold: float,
                 max_dwells: int = 1):
        self.p = folding_factor
        self.spc = samples_per_code
        self.fft_size = samples_per_code // folding_factor                                        # quicksync.cc:58
        self.n_in = samples_per_code * folding_factor
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        # wipe-off tables are n_in long, phase accumulated in float32 by the volk kernel (:84-91)
        self.wipe = np.empty((self.n_bins, self.n_in), np.complex64)
        for d in range(self.n_bins):
            doppler = -doppler_max + doppler_step * d
            phase_step = TWO_PI * np.float32(doppler) / np.float32(fs_in)
            out = np.empty(2 * self.n_in, np.float32)
            ph = C.c_float(0.0)
            lib().oracle_sincos(out, float(-phase_step), C.byref(ph), self.n_in)
            self.wipe[d] = out.view(np.complex64)
        self.init()

    def set_local_code(self, code: np.ndarray):                                                  # quicksync.cc:133-156
        self.code = np.asarray(code[:self.spc], np.complex64).copy()
        folded = np.zeros(self.fft_size, np.complex64)
        for i in range(self.p):
            folded = (folded + self.code[i * self.fft_size:(i + 1) * self.fft_size]).astype(np.complex64)
        self.code_folded = folded
        self.fft_codes = np.conj(fft_fwd(folded)).astype(np.complex64)

    def init(self):                                                                              # :180-192 (state 0)
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def _argmax(self, row):
        t = np.zeros(1, np.uint32)
        lib().oracle_index_max(t, np.ascontiguousarray(row, np.float32), len(row))
        return int(t[0])

    def work(self, x: np.ndarray) -> int:                                                        # :195-360 (state 1)
        x = np.asarray(x[:self.n_in], np.complex64)
        N = self.fft_size
        fnf = np.float32(N) * np.float32(N)
        self.mag = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.well_count += 1
        self.input_power = mean_input_power(x)                                                    # :227-230
        self.rows = []
        for d in range(self.n_bins):
            doppler = -self.doppler_max + self.doppler_step * d
            in_temp = cmul(x, self.wipe[d])                                                         # :251-253
            folded = np.zeros(N, np.complex64)
            for i in range(self.p * self.p):                                                      # :258-265
                folded = (folded + in_temp[i * N:(i + 1) * N]).astype(np.complex64)
            A = fft_fwd(folded)
            B = cmul(A, self.fft_codes)                                                           # :274-275
            y = fft_rev(B)
            mag = (y.real * y.real + y.imag * y.imag).astype(np.float32)                          # :282-283
            t = self._argmax(mag)
            magt = np.float32(mag[t] / (fnf * fnf))                                               # :289
            self.rows.append((float(magt), t))
            if self.mag < magt:                                                                   # :292
                self.mag = magt
                folded_delay = t % self.spc                                                       # :306
                possible = [folded_delay + i * N for i in range(self.p)]                          # :309-312
                acc = np.zeros(self.p, np.complex64)
                for i in range(self.p):                                                           # :314-330: sequential float sums
                    seg = cmul(in_temp[possible[i]:possible[i] + self.spc], self.code)
                    re = np.cumsum(seg.real, dtype=np.float32)[-1]
                    im = np.cumsum(seg.imag, dtype=np.float32)[-1]
                    acc[i] = np.complex64(complex(re, im))
                corr = (acc.real * acc.real + acc.imag * acc.imag).astype(np.float32)             # :332
                k = self._argmax(corr)
                self.candidates = acc
                self.result = dict(acq_delay_samples=float(possible[k]), doppler_hz=float(doppler), doppler_step=self.doppler_step,
                                   index_time=t, index_doppler=d, alias=k)
                self.test_statistics = np.float32(self.mag / self.input_power)                    # :342
        if self.test_statistics > self.threshold:                                                 # :366-375
            self.state = 2
        elif self.well_count == self.max_dwells:
            self.state = 3
        return self.state


class FineDopplerOracle:
    """pcps_acquisition_fine_doppler_cc (gnuradio_blocks/pcps_acquisition_fine_doppler_cc.cc, "fd.cc"): max_dwells 1 ms blocks are
    accumulated non-coherently into a Doppler x delay grid (:266-305), the first-to-second-peak ratio decides (:182-251), and on success
    ten code periods of signal are code-wiped and transformed with eightfold zero padding to refine the Doppler (:316-389).

    Two things are reproduced as the file has them: the wipe-off frequency of bin i is doppler_step * i - doppler_step (:170) although
    the reported Doppler is i * doppler_step - doppler_max (:243), and the replica alignment rotates only the first fft_size - 1
    samples (:339, the `- 1` in the last iterator)."""

    def __init__(self, fs_in: int, samples_per_ms: float, doppler_max: int, doppler_step: int, threshold: float, max_dwells: int,
                 consistent_grid: bool = False):
        """consistent_grid=True wipes bin i off at i * doppler_step - doppler_max, the Doppler :243 reports for it (what :170 evidently
        means); False follows :170 to the letter."""
        self.fs_in = fs_in
        self.fft_size = int(samples_per_ms)                                                      # fd.cc:61
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.n_points = int(math.floor(abs(2 * doppler_max) / doppler_step))                      # :58
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        spc = int(math.ceil((1.0 / 1.023e6) * float(np.float32(fs_in))))                          # :214
        # the grid the block actually searches: bin i at doppler_step * i - doppler_step (:170)
        self.p = PcpsOracle(fs_in, self.fft_size, doppler_max if consistent_grid else doppler_step, doppler_step, spc, float(self.fft_size),
                            num_doppler_bins=self.n_points, use_cfar=False)
        self.init()

    def set_local_code(self, code: np.ndarray):                                                  # :130-136
        self.code = np.asarray(code[:self.fft_size], np.complex64).copy()
        self.p.set_local_code(self.code)

    def init(self):                                                                              # state 0, :438-448
        self.well_count = 0
        self.test_statistics = np.float32(0.0)
        self.buffer = []
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)
        self.state = 1

    def dwell(self, x: np.ndarray) -> int:                                                       # state 1, :449-459
        x = np.asarray(x[:self.fft_size], np.complex64)
        self.well_count += 1
        self.p.doppler_grid(x, self.well_count)                                                   # :266-305 (accumulates)
        self.buffer.append(x.copy())
        if self.well_count >= self.max_dwells:
            self.state = 2
        return self.state

    def decide(self) -> int:                                                                     # state 2, :460-472 + compute_CAF :182-251
        st = self.p.statistics(self.well_count)
        self.test_statistics = np.float32(st["test_statistics"])
        self.result = dict(acq_delay_samples=float(st["index_time"]), doppler_hz=float(st["index_doppler"] * self.doppler_step - self.doppler_max),
                           doppler_step=self.doppler_step, index_time=st["index_time"], index_doppler=st["index_doppler"])
        self.state = 3 if self.test_statistics > self.threshold else 5
        return self.state

    def estimate_doppler(self, more: np.ndarray) -> int:                                         # state 3, :473-489 + :316-389
        """`more`: the samples that follow the dwell blocks (the block keeps copying input until it holds 10 ms)."""
        buf = np.concatenate(self.buffer + [np.asarray(more, np.complex64)])[:10 * self.fft_size]
        n, N = 10 * self.fft_size, self.fft_size
        M = n * 8                                                                                # :319-323
        rep = self.code.copy()
        shift = int(self.result["acq_delay_samples"])
        if shift != 0:                                                                           # :337-340: std::rotate over [0, N - 1)
            head = rep[:N - 1]
            rep[:N - 1] = np.roll(head, -((N - shift) % (N - 1)))
        rep = np.tile(rep, 10)
        z = np.zeros(M, np.complex64)
        z[:n] = cmul(buf, rep)                                                                   # :348
        Z = fft_fwd(z)
        mag = (Z.real * Z.real + Z.imag * Z.imag).astype(np.float32)
        t = np.zeros(1, np.uint32)
        lib().oracle_index_max(t, np.ascontiguousarray(mag), M)
        k = int(t[0])
        self.fine_index = k
        half = np.float32(M) / np.float32(2.0)
        if k < M // 2:                                                                           # :363-373
            f = (np.float32(self.fs_in) / np.float32(2.0)) * np.float32(k) / half
        else:
            f = (-np.float32(self.fs_in) / np.float32(2.0)) * np.float32(M // 2 - (k - M // 2)) / half
        self.fine_doppler = float(np.float32(f))
        if abs(np.float32(f) - np.float32(self.result["doppler_hz"])) < 1000:                     # :376-384
            self.result["doppler_hz"] = float(np.float32(f))
        self.state = 4
        return self.state


class E5aNoncoherentIqOracle:
    """galileo_e5a_noncoherentIQ_acquisition_caf_cc (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc): set_local_code :162-222 and the
    search of state 2 :300-650, statement by statement in float32.  Per Doppler bin up to four local codes are correlated -- data (I) and
    pilot (Q) component, each as generated ("A") and with the first code period inverted ("B", coherent times above one code period) --, for
    each component the combination with the larger row maximum is kept, the two kept |.|^2 rows are added (non-coherent I + Q) and the arg-max
    of the sum is the bin's candidate; an optional triangular filter across the Doppler bins (CAF) re-decides the Doppler.

    Quirks of the block that are reproduced because they decide what it outputs:
      * :393  `magt_QB = d_magnitudeIB[indext_QB] / ...` -- the Q-B candidate is ranked by the I-B row;
      * :187-222 the B codes are written over the first `samples_per_code` samples of the FFT input buffer only; the rest of the buffer is
        whatever the previous transform left there: with both components code I-B is [-I, Q, Q] (the pilot replica was the last thing
        copied in) and Q-B is [-Q, Q, Q]; with one component I-B is [-I, I, I];
      * the CAF weights (:556-620): no abs() in the data component's first and body loops, abs() in its last loop and in the pilot's first
        and last loops; the pilot's last-loop normalisation is evaluated in double (literal 2.0), all others in float.
    Test infrastructure (pinned to the reference block itself by tests/test_pcps_oracle_pinned.py)."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: int, threshold: float, max_dwells: int,
                 sampled_ms: int, both_signal_components: bool, caf_window_hz: int = 0, zero_padding: int = 0, bit_transition_flag: bool = False):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)                               # :113-116
        self.pa = PcpsOracle(fs_in, fft_size, doppler_max, doppler_step, 1, float(samples_per_code), num_doppler_bins=self.n_bins)
        self.fft_size = fft_size
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.sampled_ms = 1 if zero_padding > 0 else sampled_ms                                   # :85-92
        self.both = bool(both_signal_components)
        self.caf_window_hz = caf_window_hz
        self.bit_transition = bit_transition_flag
        self.inbuf = np.zeros(fft_size, np.complex64)                                             # d_fft_if's input buffer (persistent)
        self.codes = {}
        self.test_statistics = np.float32(0.0)
        self.init()

    def _code_fft(self):
        return np.conj(fft_fwd(self.inbuf)).astype(np.complex64)

    def set_local_code(self, code_i: np.ndarray, code_q: np.ndarray | None = None):              # :162-222
        n, spc = self.fft_size, self.samples_per_code
        self.inbuf[:] = np.asarray(code_i[:n], np.complex64)
        self.codes = {"IA": self._code_fft()}
        if self.both:
            self.inbuf[:] = np.asarray(code_q[:n], np.complex64)
            self.codes["QA"] = self._code_fft()
        if self.sampled_ms > 1:
            self.inbuf[:spc] = np.asarray(code_i[:spc], np.complex64) * np.complex64(-1.0)         # :191-199: the first period only
            self.codes["IB"] = self._code_fft()
            if self.both:
                self.inbuf[:spc] = np.asarray(code_q[:spc], np.complex64) * np.complex64(-1.0)
                self.codes["QB"] = self._code_fft()
        self.time_codes = None

    def init(self):                                                                              # state 0, :262-274
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def _caf(self, caf_i, caf_q):                                                                # :546-631
        f32 = np.float32
        nb = self.n_bins
        half = self.caf_window_hz // (2 * self.pa.doppler_step)
        wf = f32(0.5) / f32(half)
        h = f32(half)
        out = np.zeros(nb, np.float32)
        for di in range(0, half):                                                                 # first iterations
            acc = f32(0.0)
            for i in range(0, half + di + 1):
                acc = f32(acc + f32(caf_i[i] * f32(f32(1.0) - f32(wf * f32(di - i)))))
            den = f32(f32(f32(f32(1.0) + f32(half + di)) - f32(f32(wf * h) * f32(f32(h + f32(1.0)) / f32(2.0)))) - f32(f32(f32(wf * f32(di)) * f32(f32(di) + f32(1.0))) / f32(2.0)))
            out[di] = f32(acc / den)
            if self.both:
                acc = f32(0.0)
                for i in range(0, half + di + 1):
                    acc = f32(acc + f32(caf_q[i] * f32(f32(1.0) - f32(wf * f32(abs(di - i))))))
                den = f32(f32(f32(f32(1.0) + f32(half + di)) - f32(f32(f32(wf * h) * f32(half + 1)) / f32(2.0))) - f32(f32(f32(wf * f32(di)) * f32(di + 1)) / f32(2.0)))
                out[di] = f32(out[di] + f32(acc / den))
        for di in range(half, nb - half):                                                         # body
            acc = f32(0.0)
            for i in range(di - half, di + half + 1):
                acc = f32(acc + f32(caf_i[i] * f32(f32(1.0) - f32(wf * f32(di - i)))))
            den = f32(f32(f32(1.0) + f32(f32(2.0) * h)) - f32(f32(f32(f32(f32(2.0) * wf) * h) * f32(half + 1)) / f32(2.0)))
            out[di] = f32(acc / den)
            if self.both:
                acc = f32(0.0)
                for i in range(di - half, di + half + 1):
                    acc = f32(acc + f32(caf_q[i] * f32(f32(1.0) - f32(wf * f32(di - i)))))
                out[di] = f32(out[di] + f32(acc / den))
        for di in range(max(nb - half, 0), nb):                                                   # final iterations
            acc = f32(0.0)
            for i in range(di - half, nb):
                acc = f32(acc + f32(caf_i[i] * f32(f32(1.0) - f32(wf * f32(abs(di - i))))))
            rest = f32(nb - di - 1)
            den = f32(f32(f32(f32(f32(1.0) + h) + rest) - f32(f32(wf * h) * f32(f32(h + f32(1.0)) / f32(2.0)))) - f32(f32(f32(wf * rest) * f32(nb - di)) / f32(2.0)))
            out[di] = f32(acc / den)
            if self.both:
                acc = f32(0.0)
                for i in range(di - half, nb):
                    acc = f32(acc + f32(caf_q[i] * f32(f32(1.0) - f32(wf * f32(abs(di - i))))))
                a = f32(f32(f32(1.0) + h) + rest)                                                 # float, then the double literals take over (:617)
                t1 = float(f32(f32(wf * h) * f32(half + 1.0))) / 2.0
                t2 = float(f32(f32(wf * rest) * f32(nb - di))) / 2.0
                den_q = f32((float(a) - t1) - t2)
                out[di] = f32(out[di] + f32(acc / den_q))
        return out

    def work(self, x: np.ndarray) -> int:                                                        # state 2, :300-650
        p = self.pa
        x = np.asarray(x[:self.fft_size], np.complex64)
        f32 = np.float32
        fnf = f32(self.fft_size) * f32(self.fft_size)
        div = f32(fnf * fnf)
        self.input_power = np.float32(0.0)
        self.mag = f32(0.0)
        self.well_count += 1
        self.input_power = mean_input_power(x)                                                    # :333-335
        caf_i = np.zeros(self.n_bins, np.float32)
        caf_q = np.zeros(self.n_bins, np.float32)
        self.rows = []
        for d in range(self.n_bins):
            doppler = -int(p.doppler_max) + p.doppler_step * d
            A = fft_fwd(cmul(x, p.wipe[d]))                                                       # :343-348
            mags, idx = {}, {}
            for k, codes in self.codes.items():
                y = fft_rev(cmul(A, codes))
                mags[k] = (y.real * y.real + y.imag * y.imag).astype(np.float32)
                idx[k] = p._argmax(mags[k])
            m = {k: f32(mags[k][idx[k]] / div) for k in mags}
            if "QB" in m:
                m["QB"] = f32(mags["IB"][idx["QB"]] / div)                                        # :393 as written
            if self.sampled_ms > 1:
                sel_i = "IA" if m["IA"] >= m["IB"] else "IB"                                      # :403
                sel_q = ("QA" if m["QA"] >= m["QB"] else "QB") if self.both else None              # :410 / :446
            else:
                sel_i, sel_q = "IA", ("QA" if self.both else None)
            caf_i[d] = mags[sel_i][idx[sel_i]]
            row = mags[sel_i]
            if sel_q is not None:
                caf_q[d] = mags[sel_q][idx[sel_q]]
                row = (mags[sel_i] + mags[sel_q]).astype(np.float32)                              # :419-431
            t = p._argmax(row)
            magt = f32(row[t] / div)                                                              # :436 / :466 / :491
            self.rows.append((sel_i, sel_q, float(row[t]), t))
            if self.mag < magt:                                                                   # :496 strict
                self.mag = magt
                if self.test_statistics < f32(self.mag / self.input_power) or not self.bit_transition:    # :506
                    self.result = dict(acq_delay_samples=float(t % self.samples_per_code), doppler_hz=float(doppler),
                                       doppler_step=p.doppler_step, index_time=t, index_doppler=d)
                    self.test_statistics = f32(self.mag / self.input_power)                       # :513
        if self.caf_window_hz > 0:
            self.caf = self._caf(caf_i, caf_q)
            di = p._argmax(self.caf)                                                              # :634
            self.result["doppler_hz"] = float(-int(p.doppler_max) + p.doppler_step * di)
            self.result["caf_index_doppler"] = di
        self.caf_i, self.caf_q = caf_i, caf_q
        if self.well_count == self.max_dwells:                                                    # :651-665
            self.state = 3 if self.test_statistics > self.threshold else 4
        else:
            self.state = 1
        return self.state
