"""Python face of the tracking half of the C ABI (tests and bench only; the product host side is the
C++ in gnss-sdr_amd/host/).

``HipMulticorrelatorRealCodes`` mirrors the reference class ``Cpu_Multicorrelator_Real_Codes``
(src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:37-61): same method names, same argument
order and meaning, same borrowed-buffer semantics, so a parity test reads like the reference's own
``cpu_multicorrelator_real_codes_test.cc``.  ``CorrelatorBank`` is the batched form (gsh_bank_*).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import GSH_MAX_TAPS, CorrJob, check, fptr


class HipMulticorrelatorRealCodes:
    """Drop-in for Cpu_Multicorrelator_Real_Codes; every method forwards to one gsh_mcorr_* call."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_mcorr_create(device, C.byref(self._h)))
        self._keep = {}

    def close(self):
        if self._h:
            self._lib.gsh_mcorr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # bool init(int max_signal_length_samples, int n_correlators)                      mcorr.cc:36-50
    def init(self, max_signal_length_samples: int, n_correlators: int) -> bool:
        check(self._lib.gsh_mcorr_init(self._h, max_signal_length_samples, n_correlators))
        return True

    # bool set_local_code_and_taps(int code_length_chips, const float*, float* shifts)  mcorr.cc:53-63
    def set_local_code_and_taps(self, code_length_chips: int, local_code_in: np.ndarray, shifts_chips: np.ndarray) -> bool:
        assert local_code_in.dtype == np.float32 and shifts_chips.dtype == np.float32
        self._keep["code"] = local_code_in
        self._keep["shifts"] = shifts_chips  # borrowed: mutate in place between calls, like trk.cc:2132-2146
        check(self._lib.gsh_mcorr_set_local_code_and_taps(self._h, code_length_chips, fptr(local_code_in), fptr(shifts_chips)))
        return True

    # bool set_input_output_vectors(std::complex<float>* corr_out, const std::complex<float>* sig_in)  mcorr.cc:66-72
    def set_input_output_vectors(self, corr_out: np.ndarray, sig_in: np.ndarray) -> bool:
        assert corr_out.dtype == np.complex64 and sig_in.dtype == np.complex64 and sig_in.flags.c_contiguous
        self._keep["out"] = corr_out
        self._keep["in"] = sig_in
        check(self._lib.gsh_mcorr_set_input_output_vectors(self._h, fptr(corr_out), fptr(sig_in)))
        return True

    # void set_high_dynamics_resampler(bool)                                           mcorr.cc:163-167
    def set_high_dynamics_resampler(self, use_high_dynamics_resampler: bool) -> None:
        check(self._lib.gsh_mcorr_set_high_dynamics_resampler(self._h, int(bool(use_high_dynamics_resampler))))

    # bool Carrier_wipeoff_multicorrelator_resampler(...)  7-arg form mcorr.cc:103-126, 6-arg form :129-144
    def Carrier_wipeoff_multicorrelator_resampler(self, rem_carrier_phase_in_rad, phase_step_rad, *rest) -> bool:
        if len(rest) == 5:
            phase_rate_step_rad, rem_code, code_step, code_rate, n = rest
            check(self._lib.gsh_mcorr_carrier_wipeoff_multicorrelator_resampler(
                self._h, rem_carrier_phase_in_rad, phase_step_rad, phase_rate_step_rad, rem_code, code_step, code_rate, int(n)))
        elif len(rest) == 4:
            rem_code, code_step, code_rate, n = rest
            check(self._lib.gsh_mcorr_carrier_wipeoff_multicorrelator_resampler6(
                self._h, rem_carrier_phase_in_rad, phase_step_rad, rem_code, code_step, code_rate, int(n)))
        else:
            raise TypeError("expected the reference's 7- or 6-argument form")
        return True

    # bool free()                                                                       mcorr.cc:147-160
    def free(self) -> bool:
        check(self._lib.gsh_mcorr_free(self._h))
        return True


def make_jobs(rows: Iterable[dict]) -> "C.Array[CorrJob]":
    """Build a gsh_corr_job array from dicts with the struct's field names (shifts_chips: sequence)."""
    rows = list(rows)
    arr = (CorrJob * len(rows))()
    for j, r in zip(arr, rows):
        sh = list(r.get("shifts_chips", [0.0]))
        j.sample_offset = int(r.get("sample_offset", 0))
        j.n_samples = int(r["n_samples"])
        j.code_slot = int(r.get("code_slot", 0))
        j.rem_carr_phase_rad = float(r.get("rem_carr_phase_rad", 0.0))
        j.phase_step_rad = float(r.get("phase_step_rad", 0.0))
        j.phase_rate_step_rad = float(r.get("phase_rate_step_rad", 0.0))
        j.rem_code_phase_chips = float(r.get("rem_code_phase_chips", 0.0))
        j.code_phase_step_chips = float(r.get("code_phase_step_chips", 0.0))
        j.code_phase_rate_step_chips = float(r.get("code_phase_rate_step_chips", 0.0))
        j.n_taps = int(r.get("n_taps", len(sh)))
        j.high_dyn = int(r.get("high_dyn", 0))
        for t, v in enumerate(sh[:GSH_MAX_TAPS]):
            j.shifts_chips[t] = float(v)
    return arr


class CorrelatorBank:
    """gsh_bank_*: many (channel, epoch) jobs per launch over a device-resident IF stream."""

    def __init__(self, n_code_slots: int, max_code_length: int, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_bank_create(device, n_code_slots, max_code_length, C.byref(self._h)))
        self._keep = {}
        self.n_jobs = 0

    def close(self):
        if self._h:
            self._lib.gsh_bank_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_code(self, slot: int, code: np.ndarray) -> None:
        code = np.ascontiguousarray(code, np.float32)
        check(self._lib.gsh_bank_set_code(self._h, slot, fptr(code), len(code)))

    def set_stream_host(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, np.complex64)
        check(self._lib.gsh_bank_set_stream_host(self._h, fptr(x), len(x)))

    def set_stream_device(self, device_ptr: int, n_samples: int, keepalive=None) -> None:
        """Borrow device memory (e.g. a torch tensor's data_ptr()); the caller keeps it alive."""
        self._keep["stream"] = keepalive
        check(self._lib.gsh_bank_set_stream_device(self._h, C.c_void_p(device_ptr), n_samples))

    def set_stream_ring(self, ring) -> None:
        """Bind to a SampleStream: job sample_offset becomes an absolute sample index (None detaches)."""
        self._keep["stream"] = ring
        check(self._lib.gsh_bank_set_stream_ring(self._h, ring._h if ring is not None else None))

    def set_pair_fusion(self, enable: bool) -> None:
        """Fuse a single-tap job into the job in front of it when both read the same window with the same NCO (default on)."""
        check(self._lib.gsh_bank_set_pair_fusion(self._h, int(bool(enable))))

    def set_sample_base(self, sample_base: int) -> None:
        """Offset added to every job's sample_offset at launch: one resident job table serves block after block."""
        check(self._lib.gsh_bank_set_sample_base(self._h, int(sample_base)))

    def set_splits(self, splits: int) -> None:
        check(self._lib.gsh_bank_set_splits(self._h, splits))

    def correlate(self, jobs) -> np.ndarray:
        """Synchronous batch.  Returns complex64 [n_jobs, GSH_MAX_TAPS]."""
        if not isinstance(jobs, C.Array):
            jobs = make_jobs(jobs)
        n = len(jobs)
        out = np.zeros((n, GSH_MAX_TAPS), np.complex64)
        check(self._lib.gsh_bank_correlate(self._h, jobs, n, fptr(out)))
        self.n_jobs = n
        return out

    def upload_jobs(self, jobs) -> None:
        if not isinstance(jobs, C.Array):
            jobs = make_jobs(jobs)
        check(self._lib.gsh_bank_upload_jobs(self._h, jobs, len(jobs)))
        self.n_jobs = len(jobs)

    def launch(self, hip_stream: int = 0) -> None:
        check(self._lib.gsh_bank_launch(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self) -> None:
        check(self._lib.gsh_bank_synchronize(self._h))

    def read_outputs(self, n_jobs: int | None = None) -> np.ndarray:
        n = self.n_jobs if n_jobs is None else n_jobs
        out = np.zeros((n, GSH_MAX_TAPS), np.complex64)
        check(self._lib.gsh_bank_read_outputs(self._h, fptr(out), n))
        return out

    def time_launches(self, reps: int) -> float:
        """Average kernel milliseconds per launch (HIP events on the bank's stream)."""
        ms = C.c_float(0.0)
        check(self._lib.gsh_bank_time_launches(self._h, reps, C.byref(ms)))
        return ms.value
