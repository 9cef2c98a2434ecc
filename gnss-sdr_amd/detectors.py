"""The other PCPS detectors of the reference on the same device engine, for tests: host-side mirrors of

  pcps_tong_acquisition_cc          src/algorithms/acquisition/gnuradio_blocks/pcps_tong_acquisition_cc.cc  (tong.cc)
  galileo_pcps_8ms_acquisition_cc   src/algorithms/acquisition/gnuradio_blocks/galileo_pcps_8ms_acquisition_cc.cc  (8ms.cc)

Only the per-block state machine lives here (counters, thresholds); every sample-rate operation -- wipe-off, transforms,
|.|^2, weighting, grid accumulation, arg-max, input power -- runs on the GPU through the C ABI (gsh_acq_*).  The C++ product
classes with the same logic are host/hip_pcps_detectors.{h,cc}.
"""
from __future__ import annotations

import math

import numpy as np

from .acquisition import PcpsAcquisitionBank


def count_doppler_bins(doppler_max: int, doppler_step: int) -> int:
    """tong.cc:97-100, 8ms.cc:65-68: -doppler_max ... +doppler_max inclusive."""
    return 2 * doppler_max // doppler_step + 1 if doppler_max >= 0 else 0


def threshold_compute_doppler(pfa: float, vector_length: int, doppler_max: int, doppler_step: int) -> float:
    """ThresholdComputeDoppler::calculate_threshold (base_pcps_acquisition_custom.cc:89-112): quantile of an exponential
    distribution with lambda = vector_length at (1 - pfa)^(1 / ncells)."""
    bins = count_doppler_bins(doppler_max, doppler_step)
    ncells = vector_length * bins
    val = math.pow(1.0 - pfa, 1.0 / float(ncells))
    return float(np.float32(-math.log1p(-val) / float(vector_length)))


class PcpsTongAcquisition:
    """general_work of pcps_tong_acquisition_cc for one channel (tong.cc:147-400)."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 tong_init_val: int, tong_max_val: int, tong_max_dwells: int, device: int = 0, transform_path: int = 0):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.bank = PcpsAcquisitionBank(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, max_prn=1,
                                        num_doppler_bins=self.n_bins, device=device, transform_path=transform_path)
        self.fft_size = fft_size
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.tong_init_val, self.tong_max_val, self.tong_max_dwells = tong_init_val, tong_max_val, tong_max_dwells
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code: np.ndarray) -> None:                                          # tong.cc:136-144
        self.bank.set_local_code(0, code)

    def init(self) -> None:                                                                      # tong.cc:162-184 (state 0)
        self.dwell_count = 0
        self.tong_count = self.tong_init_val
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # tong.cc:187-301 (state 1)
        x = np.ascontiguousarray(x[:self.fft_size], np.complex64)
        self.dwell_count += 1
        # this dwell's weight needs this block's power before its magnitudes reach the grid: stage, reduce, then dwell
        self.input_power = np.float32(self.bank.stage_and_input_power(x))                         # :208-210
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)                               # :195
        self.weight = np.float32(1.0) / (fnf * fnf * self.input_power)                            # :245
        self.bank.set_grid_weight(float(self.weight))
        r = self.bank.dwell_resident(1, accumulate=self.dwell_count > 1, dwell_count=self.dwell_count)[0]
        self.mag = np.float32(r["peak"])                                                          # :252-259 (d_mag starts at 0)
        if self.mag > 0.0:
            self.result = dict(acq_delay_samples=float(r["index_time"] % self.samples_per_code),
                               doppler_hz=float(-self.doppler_max + self.doppler_step * r["index_doppler"]),
                               doppler_step=self.doppler_step, index_time=r["index_time"], index_doppler=r["index_doppler"])
        self.test_statistics = self.mag                                                           # :277
        if self.test_statistics > self.threshold * np.float32(self.dwell_count):                  # :279-294
            self.tong_count += 1
            if self.tong_count == self.tong_max_val:
                self.state = 2
        else:
            self.tong_count -= 1
            if self.tong_count == 0:
                self.state = 3
        if self.dwell_count >= self.tong_max_dwells:                                              # :296-299
            self.state = 3
        return self.state


class GalileoPcps8msAcquisition:
    """general_work of galileo_pcps_8ms_acquisition_cc for one channel (8ms.cc:134-300): local code A in slot 0, B in slot 1,
    both searched by ONE dwell over shared forward transforms."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 max_dwells: int, device: int = 0, transform_path: int = 0):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.bank = PcpsAcquisitionBank(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, max_prn=2,
                                        num_doppler_bins=self.n_bins, device=device, transform_path=transform_path)
        self.fft_size = fft_size
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code: np.ndarray) -> None:                                          # 8ms.cc:103-131
        code = np.ascontiguousarray(code[:self.fft_size], np.complex64)
        self.bank.set_local_code(0, code)
        b = code.copy()
        spc = self.samples_per_code
        b[spc:2 * spc] *= np.complex64(-1.0)
        self.bank.set_local_code(1, b)

    def init(self) -> None:                                                                      # 8ms.cc:147-160
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # 8ms.cc:163-287
        x = np.ascontiguousarray(x[:self.fft_size], np.complex64)
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)
        self.mag = np.float32(0.0)
        self.well_count += 1
        self.bank.dwell(x, 2)
        self.input_power = np.float32(self.bank.input_power())                                    # :190-192
        pa, ta = self.bank.read_row_peaks(0)
        pb, tb = self.bank.read_row_peaks(1)
        self.rows = []
        for d in range(self.n_bins):
            ma = np.float32(pa[d] / (fnf * fnf))                                                  # :222
            mb = np.float32(pb[d] / (fnf * fnf))                                                  # :237
            magt, t, which = (ma, int(ta[d]), 0) if ma >= mb else (mb, int(tb[d]), 1)             # :240-249
            self.rows.append((float(ma), int(ta[d]), float(mb), int(tb[d])))
            if self.mag < magt:                                                                   # :252
                self.mag = magt
                self.result = dict(acq_delay_samples=float(t % self.samples_per_code),
                                   doppler_hz=float(-self.doppler_max + self.doppler_step * d), doppler_step=self.doppler_step,
                                   index_time=t, index_doppler=d, code=which)
        self.test_statistics = np.float32(self.mag / self.input_power)                            # :278
        if self.test_statistics > self.threshold:
            self.state = 2
        elif self.well_count == self.max_dwells:
            self.state = 3
        return self.state


class PcpsCccwsrAcquisition:
    """general_work of pcps_cccwsr_acquisition_cc for one channel (pcps_cccwsr_acquisition_cc.cc:137-373, cited as cccwsr.cc):
    coherent channel combining with sign recovery.  The block combines the data and pilot correlations as data + j*pilot and
    data - j*pilot (cccwsr.cc:235-244).  The correlation is linear in the local code, so those two are the correlations with the
    local codes (data - j*pilot) and (data + j*pilot): slot 0 and slot 1 of ONE dwell over shared forward transforms -- the same
    one forward + two inverse transforms per bin as the block, without its two complex correlation vectors and the element-wise
    combination pass over them (|.|^2 and arg-max of each branch are fused into the cell kernel)."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: float, threshold: float,
                 max_dwells: int, device: int = 0, transform_path: int = 0):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)                               # cccwsr.cc:79-82
        self.bank = PcpsAcquisitionBank(fs_in, fft_size, doppler_max, doppler_step, 1, samples_per_code, max_prn=2,
                                        num_doppler_bins=self.n_bins, device=device, transform_path=transform_path)
        self.fft_size = fft_size
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code_data: np.ndarray, code_pilot: np.ndarray) -> None:             # cccwsr.cc:116-134
        d = np.ascontiguousarray(code_data[:self.fft_size], np.complex64)
        q = np.ascontiguousarray(code_pilot[:self.fft_size], np.complex64)
        jq = (np.complex64(1j) * q).astype(np.complex64)
        self.bank.set_local_code(0, (d - jq).astype(np.complex64))                                # -> data + j*pilot correlation
        self.bank.set_local_code(1, (d + jq).astype(np.complex64))                                # -> data - j*pilot correlation

    def init(self) -> None:                                                                      # cccwsr.cc:152-164 (state 0)
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # cccwsr.cc:166-306 (state 1)
        x = np.ascontiguousarray(x[:self.fft_size], np.complex64)
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)                               # :178
        self.well_count += 1                                                                      # :182; d_mag is NOT cleared here
        self.bank.dwell(x, 2)
        self.input_power = np.float32(self.bank.input_power())                                    # :192-194
        pp, tp = self.bank.read_row_peaks(0)
        pm, tm = self.bank.read_row_peaks(1)
        self.rows = []
        for d in range(self.n_bins):
            mp_ = np.float32(pp[d] / (fnf * fnf))                                                 # :248
            mm = np.float32(pm[d] / (fnf * fnf))                                                  # :252
            magt, t, which = (mp_, int(tp[d]), 0) if mp_ >= mm else (mm, int(tm[d]), 1)           # :254-263
            self.rows.append((float(mp_), int(tp[d]), float(mm), int(tm[d])))
            if self.mag < magt:                                                                   # :266
                self.mag = magt
                self.result = dict(acq_delay_samples=float(t % self.samples_per_code),
                                   doppler_hz=float(-self.doppler_max + self.doppler_step * d), doppler_step=self.doppler_step,
                                   index_time=t, index_doppler=d, branch=which)
        self.test_statistics = np.float32(self.mag / self.input_power)                            # :292
        if self.test_statistics > self.threshold:                                                 # :295
            self.state = 2
        elif self.well_count == self.max_dwells:                                                  # :299
            self.state = 3
        return self.state


class PcpsQuickSyncAcquisition:
    """general_work of pcps_quicksync_acquisition_cc for one channel (pcps_quicksync_acquisition_cc.cc:155-400, "qs.cc").
    The handle is created with fold = folding_factor^2: wipe-off, folding, both transforms, |.|^2 and the per-bin maxima run in
    one dwell; the alias of the winning folded delay is resolved by gsh_acq_time_correlate on the resident block."""

    def __init__(self, fs_in: int, samples_per_code: int, folding_factor: int, doppler_max: int, doppler_step: int, threshold: float,
                 max_dwells: int = 1, bit_transition_flag: bool = False, device: int = 0, transform_path: int = 0):
        self.p = int(folding_factor)
        self.spc = int(samples_per_code)
        self.fft_size = self.spc // self.p                                                        # qs.cc:58
        self.n_in = self.spc * self.p
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)
        self.bank = PcpsAcquisitionBank(fs_in, self.fft_size, doppler_max, doppler_step, 1, float(self.spc), max_prn=1,
                                        num_doppler_bins=self.n_bins, consumed_samples=self.n_in, fold=self.p * self.p, device=device,
                                        transform_path=transform_path)
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.threshold = np.float32(threshold)
        self.bit_transition_flag = bool(bit_transition_flag)
        self.max_dwells = 2 if self.bit_transition_flag else max_dwells                            # adapter, gps_l1_ca_pcps_quicksync_acquisition.cc:74
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code: np.ndarray) -> None:                                          # qs.cc:133-156
        self.code = np.ascontiguousarray(code[:self.spc], np.complex64).copy()
        folded = np.zeros(self.fft_size, np.complex64)
        for i in range(self.p):
            folded = (folded + self.code[i * self.fft_size:(i + 1) * self.fft_size]).astype(np.complex64)
        self.bank.set_local_code(0, folded)

    def init(self) -> None:                                                                      # qs.cc:180-192
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def work(self, x: np.ndarray) -> int:                                                        # qs.cc:195-395
        x = np.ascontiguousarray(x[:self.n_in], np.complex64)
        fnf = np.float32(self.fft_size) * np.float32(self.fft_size)
        self.mag = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.well_count += 1
        self.bank.dwell(x, 1)
        self.input_power = np.float32(self.bank.input_power())                                    # :227-230
        pk, ix = self.bank.read_row_peaks(0)
        self.rows = []
        best = None
        for d in range(self.n_bins):
            magt = np.float32(pk[d] / (fnf * fnf))                                                # :289
            self.rows.append((float(magt), int(ix[d])))
            if self.mag < magt:                                                                   # :292
                self.mag = magt
                best = (d, int(ix[d]))
        if best is not None:
            # :306-343 runs at every update of d_mag; only the last one (the winning bin) survives, so it is done once
            d, t = best
            possible = [t % self.spc + i * self.fft_size for i in range(self.p)]
            acc = self.bank.time_correlate(self.code, d, possible)
            corr = (acc.real * acc.real + acc.imag * acc.imag).astype(np.float32)
            k = int(np.argmax(corr))
            self.candidates = acc
            self.result = dict(acq_delay_samples=float(possible[k]), doppler_hz=float(-self.doppler_max + self.doppler_step * d),
                               doppler_step=self.doppler_step, index_time=t, index_doppler=d, alias=k)
            self.test_statistics = np.float32(self.mag / self.input_power)                        # :342
        if not self.bit_transition_flag:                                                          # :366-375
            if self.test_statistics > self.threshold:
                self.state = 2
            elif self.well_count == self.max_dwells:
                self.state = 3
        elif self.well_count == self.max_dwells:                                                  # :377-390
            self.state = 2 if self.test_statistics > self.threshold else 3
        return self.state


class PcpsAcquisitionFineDoppler:
    """pcps_acquisition_fine_doppler_cc for one channel (gnuradio_blocks/pcps_acquisition_fine_doppler_cc.cc, "fd.cc"): non-coherent
    accumulation of max_dwells 1 ms grids and the peak-ratio statistic are the engine's accumulate path (gsh_acq_dwell with use_cfar = 0);
    the fine-Doppler step is gsh_spectrum_peak.  consistent_grid: see FineDopplerOracle / host/hip_pcps_detectors.h."""

    def __init__(self, fs_in: int, samples_per_ms: float, doppler_max: int, doppler_step: int, threshold: float, max_dwells: int,
                 consistent_grid: bool = False, device: int = 0, transform_path: int = 0):
        self.fs_in = fs_in
        self.fft_size = int(samples_per_ms)
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.n_points = int(math.floor(abs(2 * doppler_max) / doppler_step))
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.device = device
        spc = int(math.ceil((1.0 / 1.023e6) * float(np.float32(fs_in))))
        self.bank = PcpsAcquisitionBank(fs_in, self.fft_size, doppler_max if consistent_grid else doppler_step, doppler_step, spc, float(self.fft_size),
                                        max_prn=1, num_doppler_bins=self.n_points, use_cfar=False, device=device, transform_path=transform_path)
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code: np.ndarray) -> None:                                          # fd.cc:130-136
        self.code = np.ascontiguousarray(code[:self.fft_size], np.complex64).copy()
        self.bank.set_local_code(0, self.code)

    def init(self) -> None:                                                                      # state 0
        self.well_count = 0
        self.test_statistics = np.float32(0.0)
        self.buffer = []
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)
        self.state = 1

    def dwell(self, x: np.ndarray) -> int:                                                       # state 1
        x = np.ascontiguousarray(x[:self.fft_size], np.complex64)
        self.well_count += 1
        self._last = self.bank.dwell(x, 1, accumulate=self.well_count > 1, dwell_count=self.well_count)[0]
        self.buffer.append(x.copy())
        if self.well_count >= self.max_dwells:
            self.state = 2
        return self.state

    def decide(self) -> int:                                                                     # state 2 + compute_CAF
        r = self._last
        self.test_statistics = np.float32(r["test_statistics"])
        self.result = dict(acq_delay_samples=float(r["index_time"]), doppler_hz=float(r["index_doppler"] * self.doppler_step - self.doppler_max),
                           doppler_step=self.doppler_step, index_time=r["index_time"], index_doppler=r["index_doppler"])
        self.state = 3 if self.test_statistics > self.threshold else 5
        return self.state

    def estimate_doppler(self, more: np.ndarray) -> int:                                         # state 3 + estimate_Doppler
        import ctypes as C
        from . import _lib
        from ._lib import check, fptr
        buf = np.ascontiguousarray(np.concatenate(self.buffer + [np.asarray(more, np.complex64)])[:10 * self.fft_size])
        n, N = 10 * self.fft_size, self.fft_size
        M = n * 8
        rep = self.code.copy()
        shift = int(self.result["acq_delay_samples"])
        if shift != 0:
            rep[:N - 1] = np.roll(rep[:N - 1], -((N - shift) % (N - 1)))                          # std::rotate over [0, N - 1), fd.cc:337-340
        rep = np.ascontiguousarray(np.tile(rep, 10))
        k = C.c_uint32(0)
        pk = C.c_float(0.0)
        check(_lib.load().gsh_spectrum_peak(self.device, fptr(buf), fptr(rep), n, M, C.byref(k), C.byref(pk)))
        k = int(k.value)
        self.fine_index = k
        half = np.float32(M) / np.float32(2.0)
        if k < M // 2:
            f = (np.float32(self.fs_in) / np.float32(2.0)) * np.float32(k) / half
        else:
            f = (-np.float32(self.fs_in) / np.float32(2.0)) * np.float32(M // 2 - (k - M // 2)) / half
        self.fine_doppler = float(np.float32(f))
        if abs(np.float32(f) - np.float32(self.result["doppler_hz"])) < 1000:
            self.result["doppler_hz"] = float(np.float32(f))
        self.state = 4
        return self.state


class GalileoE5aNoncoherentIQAcquisitionCaf:
    """The search of galileo_e5a_noncoherentIQ_acquisition_caf_cc (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc: set_local_code :162-222, state 2
    :300-665) for one channel: the block's local codes I-A, Q-A, I-B, Q-B in up to four slots of ONE dwell over shared forward transforms, the
    per-bin A / B choice, the I + Q addition of the kept magnitude rows and the arg-max of the sum on the device
    (gsh_acq_noncoherent_pair_peaks), the bin loop, the statistic and the CAF filter here.  The C++ product class is
    Hip_Galileo_E5a_Noncoherent_Iq_Core (host/hip_pcps_detectors.{h,cc})."""

    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_code: int, threshold: float, max_dwells: int,
                 sampled_ms: int, both_signal_components: bool, caf_window_hz: int = 0, zero_padding: int = 0, bit_transition_flag: bool = False,
                 device: int = 0, transform_path: int = 0):
        self.n_bins = count_doppler_bins(doppler_max, doppler_step)                               # :113-116
        self.sampled_ms = 1 if zero_padding > 0 else sampled_ms                                   # :85-92
        self.both = bool(both_signal_components)
        self.slots = {"IA": 0}
        if self.both:
            self.slots["QA"] = len(self.slots)
        if self.sampled_ms > 1:
            self.slots["IB"] = len(self.slots)
            if self.both:
                self.slots["QB"] = len(self.slots)
        self.bank = PcpsAcquisitionBank(fs_in, fft_size, doppler_max, doppler_step, 1, float(samples_per_code), max_prn=len(self.slots),
                                        num_doppler_bins=self.n_bins, device=device, transform_path=transform_path)
        self.fft_size = fft_size
        self.doppler_max, self.doppler_step = doppler_max, doppler_step
        self.samples_per_code = int(samples_per_code)
        self.threshold = np.float32(threshold)
        self.max_dwells = max_dwells
        self.caf_window_hz = caf_window_hz
        self.bit_transition = bit_transition_flag
        self.inbuf = np.zeros(fft_size, np.complex64)   # the block's FFT input buffer: the B codes overwrite its first code period only (:187-222)
        self.test_statistics = np.float32(0.0)
        self.init()

    def close(self):
        self.bank.close()

    def set_local_code(self, code_i: np.ndarray, code_q: np.ndarray | None = None) -> None:      # :162-222
        n, spc = self.fft_size, self.samples_per_code
        self.inbuf[:] = np.asarray(code_i[:n], np.complex64)
        self.bank.set_local_code(self.slots["IA"], self.inbuf)
        if self.both:
            self.inbuf[:] = np.asarray(code_q[:n], np.complex64)
            self.bank.set_local_code(self.slots["QA"], self.inbuf)
        if self.sampled_ms > 1:
            self.inbuf[:spc] = np.asarray(code_i[:spc], np.complex64) * np.complex64(-1.0)
            self.bank.set_local_code(self.slots["IB"], self.inbuf)
            if self.both:
                self.inbuf[:spc] = np.asarray(code_q[:spc], np.complex64) * np.complex64(-1.0)
                self.bank.set_local_code(self.slots["QB"], self.inbuf)

    def init(self) -> None:                                                                      # state 0, :262-274
        self.well_count = 0
        self.mag = np.float32(0.0)
        self.input_power = np.float32(0.0)
        self.test_statistics = np.float32(0.0)
        self.state = 1
        self.result = dict(acq_delay_samples=0.0, doppler_hz=0.0, doppler_step=0)

    def _caf(self, caf_i, caf_q):
        """:546-631, in the block's own float / double mix (see oracle.pcps_oracle.E5aNoncoherentIqOracle for the statement-by-statement notes)."""
        f32 = np.float32
        nb = self.n_bins
        half = self.caf_window_hz // (2 * self.doppler_step)
        wf = f32(0.5) / f32(half)
        h = f32(half)
        out = np.zeros(nb, np.float32)

        def tri(vec, lo, hi, di, use_abs):
            acc = f32(0.0)
            for i in range(lo, hi):
                k = abs(di - i) if use_abs else di - i
                acc = f32(acc + f32(vec[i] * f32(f32(1.0) - f32(wf * f32(k)))))
            return acc

        for di in range(0, half):
            den = f32(f32(f32(f32(1.0) + f32(half + di)) - f32(f32(wf * h) * f32(f32(h + f32(1.0)) / f32(2.0)))) - f32(f32(f32(wf * f32(di)) * f32(f32(di) + f32(1.0))) / f32(2.0)))
            out[di] = f32(tri(caf_i, 0, half + di + 1, di, False) / den)
            if self.both:
                den = f32(f32(f32(f32(1.0) + f32(half + di)) - f32(f32(f32(wf * h) * f32(half + 1)) / f32(2.0))) - f32(f32(f32(wf * f32(di)) * f32(di + 1)) / f32(2.0)))
                out[di] = f32(out[di] + f32(tri(caf_q, 0, half + di + 1, di, True) / den))
        for di in range(half, nb - half):
            den = f32(f32(f32(1.0) + f32(f32(2.0) * h)) - f32(f32(f32(f32(f32(2.0) * wf) * h) * f32(half + 1)) / f32(2.0)))
            out[di] = f32(tri(caf_i, di - half, di + half + 1, di, False) / den)
            if self.both:
                out[di] = f32(out[di] + f32(tri(caf_q, di - half, di + half + 1, di, False) / den))
        for di in range(max(nb - half, 0), nb):
            rest = f32(nb - di - 1)
            den = f32(f32(f32(f32(f32(1.0) + h) + rest) - f32(f32(wf * h) * f32(f32(h + f32(1.0)) / f32(2.0)))) - f32(f32(f32(wf * rest) * f32(nb - di)) / f32(2.0)))
            out[di] = f32(tri(caf_i, di - half, nb, di, True) / den)
            if self.both:
                a = f32(f32(f32(1.0) + h) + rest)
                t1 = float(f32(f32(wf * h) * f32(half + 1.0))) / 2.0
                t2 = float(f32(f32(wf * rest) * f32(nb - di))) / 2.0
                out[di] = f32(out[di] + f32(tri(caf_q, di - half, nb, di, True) / f32((float(a) - t1) - t2)))
        return out

    def work(self, x: np.ndarray) -> int:                                                        # state 2, :300-665
        x = np.ascontiguousarray(x[:self.fft_size], np.complex64)
        f32 = np.float32
        fnf = f32(self.fft_size) * f32(self.fft_size)
        div = f32(fnf * fnf)
        self.mag = f32(0.0)
        self.well_count += 1
        self.bank.dwell(x, len(self.slots))
        self.input_power = f32(self.bank.input_power())                                           # :333-335
        s = self.slots
        pk = self.bank.noncoherent_pair_peaks(s["IA"], s.get("QA", -1), s.get("IB", -1), s.get("QB", -1))
        self.rows = pk
        for d in range(self.n_bins):
            magt = f32(pk["peak"][d] / div)                                                       # :436 / :466 / :491
            t = int(pk["index_time"][d])
            if self.mag < magt:                                                                   # :496 strict
                self.mag = magt
                if self.test_statistics < f32(self.mag / self.input_power) or not self.bit_transition:    # :506
                    self.result = dict(acq_delay_samples=float(t % self.samples_per_code), doppler_hz=float(-self.doppler_max + self.doppler_step * d),
                                       doppler_step=self.doppler_step, index_time=t, index_doppler=d)
                    self.test_statistics = f32(self.mag / self.input_power)                       # :513
        if self.caf_window_hz > 0:
            self.caf = self._caf(pk["caf_i"], pk["caf_q"])
            best = self.caf.max()
            di = int(np.nonzero(self.caf == best)[0][0])                                          # lowest index among equals (volk_gnsssdr_32f_index_max_32u)
            self.result["doppler_hz"] = float(-self.doppler_max + self.doppler_step * di)         # :634-636
            self.result["caf_index_doppler"] = di
        if self.well_count == self.max_dwells:                                                    # :651-665
            self.state = 3 if self.test_statistics > self.threshold else 4
        else:
            self.state = 1
        return self.state
