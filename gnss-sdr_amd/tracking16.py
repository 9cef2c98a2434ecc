"""Python face of the 16-bit correlator family of the C ABI (tests and bench only).

``HipMulticorrelator16sc`` mirrors the reference class ``Cpu_Multicorrelator_16sc``
(src/algorithms/tracking/libs/cpu_multicorrelator_16sc.h:38-57): same method names, argument order and borrowed-buffer
semantics.  ``CorrelatorBank16`` is the batched form (gsh_bank16_*).  Complex int16 data are numpy arrays of shape [n, 2]
(I, Q), dtype int16 -- the memory layout of lv_16sc_t.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import numpy as np

from . import _lib
from ._lib import GSH_MAX_TAPS, Corr16Job, check


def _i16(arr: np.ndarray):
    assert arr.dtype == np.int16 and arr.flags.c_contiguous
    return arr.ctypes.data_as(C.POINTER(C.c_int16))


class HipMulticorrelator16sc:
    """Drop-in for Cpu_Multicorrelator_16sc; every method forwards to one gsh_mcorr16_* call."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_mcorr16_create(device, C.byref(self._h)))
        self._keep = {}

    def close(self):
        if self._h:
            self._lib.gsh_mcorr16_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # bool init(int max_signal_length_samples, int n_correlators)                                   cpu_multicorrelator_16sc.cc:25-41
    def init(self, max_signal_length_samples: int, n_correlators: int) -> bool:
        check(self._lib.gsh_mcorr16_init(self._h, max_signal_length_samples, n_correlators))
        return True

    # bool set_local_code_and_taps(int code_length_chips, const lv_16sc_t* local_code_in, float* shifts_chips)   .cc:44-53
    def set_local_code_and_taps(self, code_length_chips: int, local_code_in: np.ndarray, shifts_chips: np.ndarray) -> bool:
        assert shifts_chips.dtype == np.float32
        self._keep["code"] = local_code_in
        self._keep["shifts"] = shifts_chips  # borrowed: may be changed in place between calls
        check(self._lib.gsh_mcorr16_set_local_code_and_taps(self._h, code_length_chips, _i16(local_code_in), shifts_chips.ctypes.data_as(C.POINTER(C.c_float))))
        return True

    # bool set_input_output_vectors(lv_16sc_t* corr_out, const lv_16sc_t* sig_in)                   .cc:56-62
    def set_input_output_vectors(self, corr_out: np.ndarray, sig_in: np.ndarray) -> bool:
        self._keep["out"] = corr_out
        self._keep["in"] = sig_in
        check(self._lib.gsh_mcorr16_set_input_output_vectors(self._h, _i16(corr_out), _i16(sig_in)))
        return True

    # bool Carrier_wipeoff_multicorrelator_resampler(float, float, float, float, int)                .cc:80-96
    def Carrier_wipeoff_multicorrelator_resampler(self, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips, code_phase_step_chips,
                                                  signal_length_samples) -> bool:
        check(self._lib.gsh_mcorr16_carrier_wipeoff_multicorrelator_resampler(self._h, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips,
                                                                                code_phase_step_chips, int(signal_length_samples)))
        return True

    # bool free()                                                                                    .cc:107-120
    def free(self) -> bool:
        check(self._lib.gsh_mcorr16_free(self._h))
        return True


def make_job16(sample_offset, n_samples, code_slot, rem_carr, phase_step, rem_code, code_step, shifts: Iterable[float]) -> Corr16Job:
    j = Corr16Job()
    j.sample_offset = int(sample_offset)
    j.n_samples = int(n_samples)
    j.code_slot = int(code_slot)
    j.rem_carr_phase_rad = rem_carr
    j.phase_step_rad = phase_step
    j.rem_code_phase_chips = rem_code
    j.code_phase_step_chips = code_step
    shifts = list(shifts)
    j.n_taps = len(shifts)
    for t, s in enumerate(shifts):
        j.shifts_chips[t] = s
    return j


class CorrelatorBank16:
    """gsh_bank16_*: many Cpu_Multicorrelator_16sc calls over one device-resident int16 stream in one launch."""

    def __init__(self, n_code_slots: int, max_code_length: int, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_bank16_create(device, n_code_slots, max_code_length, C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            self._lib.gsh_bank16_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_code(self, slot: int, code_iq: np.ndarray) -> None:
        code_iq = np.ascontiguousarray(code_iq, np.int16).reshape(-1, 2)
        check(self._lib.gsh_bank16_set_code(self._h, slot, _i16(code_iq), len(code_iq)))

    def set_stream_host(self, iq: np.ndarray) -> None:
        iq = np.ascontiguousarray(iq, np.int16).reshape(-1, 2)
        check(self._lib.gsh_bank16_set_stream_host(self._h, _i16(iq), len(iq)))

    def set_stream_device(self, ptr: int, n_samples: int, keepalive=None) -> None:
        self._keep = keepalive
        check(self._lib.gsh_bank16_set_stream_device(self._h, C.c_void_p(ptr), n_samples))

    def correlate(self, jobs) -> np.ndarray:
        """-> int16[n_jobs, GSH_MAX_TAPS, 2]"""
        arr = (Corr16Job * len(jobs))(*jobs)
        out = np.zeros((len(jobs), GSH_MAX_TAPS, 2), np.int16)
        check(self._lib.gsh_bank16_correlate(self._h, arr, len(jobs), _i16(out)))
        return out

    def upload(self, jobs) -> None:
        arr = (Corr16Job * len(jobs))(*jobs)
        check(self._lib.gsh_bank16_upload_jobs(self._h, arr, len(jobs)))
        self._n = len(jobs)

    def launch(self) -> None:
        check(self._lib.gsh_bank16_launch(self._h))

    def read(self) -> np.ndarray:
        out = np.zeros((self._n, GSH_MAX_TAPS, 2), np.int16)
        check(self._lib.gsh_bank16_read_outputs(self._h, _i16(out), self._n))
        return out

    def time_launches(self, reps: int) -> float:
        ms = C.c_float(0.0)
        check(self._lib.gsh_bank16_time_launches(self._h, reps, C.byref(ms)))
        return ms.value
