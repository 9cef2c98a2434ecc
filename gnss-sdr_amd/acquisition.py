"""Python face of the acquisition half of the C ABI (gsh_acq_*), for tests and bench.

``PcpsAcquisitionBank`` wraps one handle: the arithmetic core of the reference's ``pcps_acquisition`` block
(src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc) for up to ``max_prn`` local codes at once.
Method names follow the reference block: set_local_code (:218), set_doppler_center (:737), and ``dwell`` = one
acquisition_core pass (:648) returning what max_to_input_power_statistic / first_vs_second_peak_statistic and
update_synchro produce.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import AcqConf, AcqResult, check, fptr


def acq_sizes(fs_in: int, sampled_ms: int = 1, ms_per_code: int = 1, bit_transition_flag: bool = False, chips_per_second: float = 1.023e6):
    """Derived sizes exactly as the reference computes them (acq_conf.cc:119-124, pcps_acquisition.cc:110-112)."""
    samples_per_ms = np.float32(fs_in) * np.float32(0.001)
    consumed = int(np.float64(sampled_ms) * np.float64(samples_per_ms) * (2.0 if bit_transition_flag else 1.0))
    fft_size = consumed if sampled_ms == ms_per_code else 2 * consumed
    effective = fft_size // 2 if bit_transition_flag else fft_size
    samples_per_chip = int(math.ceil(float(np.float32(fs_in) / np.float32(chips_per_second))))
    samples_per_code = float(samples_per_ms * np.float32(ms_per_code))
    return dict(consumed_samples=consumed, fft_size=fft_size, effective_fft_size=effective,
                samples_per_chip=samples_per_chip, samples_per_code=samples_per_code)


class PcpsAcquisitionBank:
    def __init__(self, fs_in: int, fft_size: int, doppler_max: int, doppler_step: int, samples_per_chip: int,
                 samples_per_code: float, max_prn: int = 1, num_doppler_bins: int = 0, consumed_samples: int | None = None,
                 effective_fft_size: int | None = None, doppler_center: int = 0, doppler_bias: int = 0,
                 bit_transition_flag: bool = False, use_cfar: bool = True, device: int = 0, keep_grid: bool = True,
                 transform_path: int = 0, num_doppler_bins_step2: int = 0, doppler_step2: float = 125.0, fold: int = 0):
        self._lib = _lib.load()
        c = AcqConf()
        c.fs_in = int(fs_in)
        c.fft_size = int(fft_size)
        c.consumed_samples = int(fft_size if consumed_samples is None else consumed_samples)
        c.effective_fft_size = int((fft_size // 2 if bit_transition_flag else fft_size) if effective_fft_size is None else effective_fft_size)
        c.num_doppler_bins = int(num_doppler_bins)
        c.doppler_max = int(doppler_max)
        c.doppler_step = int(doppler_step)
        c.doppler_center = int(doppler_center)
        c.doppler_bias = int(doppler_bias)
        c.samples_per_chip = int(samples_per_chip)
        c.samples_per_code = float(samples_per_code)
        c.bit_transition_flag = int(bool(bit_transition_flag))
        c.use_cfar = int(bool(use_cfar))
        c.max_prn = int(max_prn)
        c.no_grid = 0 if keep_grid else 1  # keep_grid=False: max_dwells == 1 and no dump (no accumulate, no read_grid)
        c.transform_path = int(transform_path)
        c.num_doppler_bins_step2 = int(num_doppler_bins_step2)  # 0: make_two_steps off
        c.doppler_step2 = float(doppler_step2)
        c.fold = int(fold)  # > 1: QuickSync folding, consumed_samples = fold * fft_size
        self.conf = c
        self.num_doppler_bins = int(num_doppler_bins) if num_doppler_bins else int(math.ceil(2.0 * doppler_max / doppler_step))
        self._h = C.c_void_p()
        check(self._lib.gsh_acq_create(device, C.byref(c), C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gsh_acq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_local_code(self, prn_slot: int, code: np.ndarray) -> None:
        """code: the time-domain complex replica the adapter generates (e.g. gps_l1_ca_code_gen_complex_sampled),
        consumed_samples long (fft_size/2 with bit_transition_flag)."""
        code = np.ascontiguousarray(code, np.complex64)
        need = self.conf.fft_size // 2 if self.conf.bit_transition_flag else (self.conf.fft_size if self.conf.fold > 1 else self.conf.consumed_samples)
        if len(code) < need:
            raise ValueError(f"code has {len(code)} samples, {need} needed")
        check(self._lib.gsh_acq_set_local_code(self._h, prn_slot, fptr(code)))

    def set_doppler_center(self, doppler_center: int) -> None:
        check(self._lib.gsh_acq_set_doppler_center(self._h, int(doppler_center)))

    def set_doppler_bias(self, doppler_bias: int) -> None:
        """GLONASS FDMA carrier offset of the satellite being searched (acq.cc:252-272, 289)."""
        check(self._lib.gsh_acq_set_doppler_bias(self._h, int(doppler_bias)))

    def set_grid_weight(self, weight: float) -> None:
        """Factor applied to every |y|^2 before it reaches the grid (pcps_tong_acquisition_cc.cc:243-249)."""
        check(self._lib.gsh_acq_set_grid_weight(self._h, float(weight)))

    def input_power(self) -> float:
        """mean |x|^2 of the block most recently handed to a dwell (pcps_tong_acquisition_cc.cc:208-210)."""
        p = C.c_float(0.0)
        check(self._lib.gsh_acq_input_power(self._h, C.byref(p)))
        return float(p.value)

    def stage_input(self, x) -> None:
        """Make x[:consumed_samples] (numpy array or torch cuda tensor) the handle's resident block without running a dwell."""
        if hasattr(x, "data_ptr"):
            check(self._lib.gsh_acq_stage_input_device(self._h, C.c_void_p(x.data_ptr())))
            return
        x = np.ascontiguousarray(x, np.complex64)
        if len(x) < self.conf.consumed_samples:
            raise ValueError("input shorter than consumed_samples")
        check(self._lib.gsh_acq_stage_input(self._h, fptr(x)))

    def stage_and_input_power(self, x) -> float:
        self.stage_input(x)
        return self.input_power()

    def dwell_resident(self, n_prn: int, accumulate: bool = False, dwell_count: int = 1):
        res = (AcqResult * n_prn)()
        check(self._lib.gsh_acq_dwell_resident(self._h, n_prn, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def time_correlate(self, code: np.ndarray, doppler_index: int, delays) -> np.ndarray:
        """sum_j x[delay + j] w_bin[delay + j] code[j] over the resident block for each candidate delay (quicksync.cc:295-323)."""
        code = np.ascontiguousarray(code, np.complex64)
        d = np.ascontiguousarray(delays, np.uint32)
        out = np.empty(len(d), np.complex64)
        check(self._lib.gsh_acq_time_correlate(self._h, fptr(code), len(code), int(doppler_index), d.ctypes.data_as(C.POINTER(C.c_uint32)), len(d), fptr(out)))
        return out

    def read_row_peaks(self, prn_slot: int):
        """(per-bin maximum, per-bin lowest arg-max) of prn_slot's grid after the last dwell."""
        pk = np.empty(self.num_doppler_bins, np.float32)
        ix = np.empty(self.num_doppler_bins, np.uint32)
        check(self._lib.gsh_acq_read_row_peaks(self._h, prn_slot, fptr(pk), ix.ctypes.data_as(C.POINTER(C.c_uint32))))
        return pk, ix

    PAIR_PEAK = np.dtype([("peak", np.float32), ("index_time", np.uint32), ("caf_i", np.float32), ("caf_q", np.float32), ("i_slot", np.uint32), ("q_slot", np.uint32)])

    def noncoherent_pair_peaks(self, slot_ia: int, slot_qa: int = -1, slot_ib: int = -1, slot_qb: int = -1) -> np.ndarray:
        """gsh_acq_noncoherent_pair_peaks after a dwell: per Doppler bin the maximum of the SUM of the kept data and pilot magnitude rows
        (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc:357-492), added on the device.  Structured array, one record per bin."""
        out = np.zeros(self.num_doppler_bins, self.PAIR_PEAK)
        check(self._lib.gsh_acq_noncoherent_pair_peaks(self._h, slot_ia, slot_qa, slot_ib, slot_qb, out.ctypes.data))
        return out

    def dwell(self, x: np.ndarray, n_prn: int, accumulate: bool = False, dwell_count: int = 1):
        """One acquisition_core pass over x[:consumed_samples] for prn slots 0..n_prn-1.  Returns a list of dicts."""
        x = np.ascontiguousarray(x, np.complex64)
        if len(x) < self.conf.consumed_samples:
            raise ValueError("input shorter than consumed_samples")
        res = (AcqResult * n_prn)()
        check(self._lib.gsh_acq_dwell(self._h, fptr(x), n_prn, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def dwell_device(self, device_ptr: int, n_prn: int, accumulate: bool = False, dwell_count: int = 1):
        res = (AcqResult * n_prn)()
        check(self._lib.gsh_acq_dwell_device(self._h, C.c_void_p(device_ptr), n_prn, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def dwell_cshort(self, x16: np.ndarray, n_prn: int, accumulate: bool = False, dwell_count: int = 1):
        """item_type = cshort (acq.cc:653-656): x16 is int16 [consumed_samples, 2] (I, Q), converted on the device."""
        x16 = np.ascontiguousarray(x16, np.int16).reshape(-1)
        if len(x16) < 2 * self.conf.consumed_samples:
            raise ValueError("input shorter than consumed_samples")
        res = (AcqResult * n_prn)()
        check(self._lib.gsh_acq_dwell_cshort(self._h, x16.ctypes.data_as(C.POINTER(C.c_int16)), n_prn, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def dwell_step2(self, x: np.ndarray, prn_slots, doppler_centers, input_powers=None, accumulate: bool = False, dwell_count: int = 1):
        """Step two of make_two_steps (acq.cc:294-301, 428-437, 605-624): a narrow grid around each slot's step-one Doppler."""
        x = np.ascontiguousarray(x, np.complex64)
        n = len(prn_slots)
        slots = np.ascontiguousarray(prn_slots, np.uint32)
        cen = np.ascontiguousarray(doppler_centers, np.float32)
        pw = np.ascontiguousarray(input_powers if input_powers is not None else np.zeros(n), np.float32)
        res = (AcqResult * n)()
        check(self._lib.gsh_acq_dwell_step2(self._h, fptr(x), n, slots.ctypes.data_as(C.POINTER(C.c_uint32)), fptr(cen),
                                            fptr(pw) if (input_powers is not None or self.conf.use_cfar) else None, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def dwell_ring(self, ring, first_sample: int, n_prn: int, accumulate: bool = False, dwell_count: int = 1):
        """One dwell over consumed_samples resident samples of a SampleStream ring starting at absolute index first_sample."""
        res = (AcqResult * n_prn)()
        check(self._lib.gsh_acq_dwell_ring(self._h, ring._h, int(first_sample), n_prn, int(accumulate), dwell_count, res))
        return [self._to_dict(r) for r in res]

    def read_grid(self, prn_slot: int) -> np.ndarray:
        g = np.empty((self.num_doppler_bins, self.conf.effective_fft_size), np.float32)
        check(self._lib.gsh_acq_read_grid(self._h, prn_slot, fptr(g)))
        return g

    def time_dwells(self, x, n_prn: int, reps: int = 10, pipelined: bool = False) -> float:
        """Average milliseconds per full dwell batch, input resident (x: numpy array or torch cuda tensor).
        pipelined=True: the batches are issued on two HIP streams (steady-state throughput, not latency)."""
        if hasattr(x, "data_ptr"):
            self.dwell_device(x.data_ptr(), n_prn)
        else:
            self.dwell(x, n_prn)
        ms = C.c_float(0.0)
        fn = self._lib.gsh_acq_time_dwells_pipelined if pipelined else self._lib.gsh_acq_time_dwells
        check(fn(self._h, n_prn, reps, C.byref(ms)))
        return ms.value

    @staticmethod
    def _to_dict(r: AcqResult) -> dict:
        return dict(index_time=int(r.index_time), index_doppler=int(r.index_doppler), doppler_hz=int(r.doppler_hz),
                    acq_delay_samples=float(r.acq_delay_samples), peak=float(r.peak), input_power=float(r.input_power),
                    second_peak=float(r.second_peak), test_statistics=float(r.test_statistics))


def compute_threshold(pfa: float, effective_fft_size: int, num_doppler_bins: int, max_dwells: int) -> float:
    return float(_lib.load().gsh_acq_compute_threshold(pfa, effective_fft_size, num_doppler_bins, max_dwells))
