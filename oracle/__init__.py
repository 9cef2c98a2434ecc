"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the two CPU checkers.

``oracle.lib()``  -> liboracle.so, our own C restatement (oracle/gnss_oracle.c).
``oracle.ref()``  -> _ref/libgnsssdr_ref.so, the REFERENCE's own sources compiled from
                     /root/reference by oracle/Makefile (None when it was never built).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``gnss-sdr_amd/``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")



class LoopFilter(C.Structure):
    """oracle_loop_filter"""
    _fields_ = [("in_c", C.c_float * 4), ("out_c", C.c_float * 4), ("in_h", C.c_float * 4), ("out_h", C.c_float * 4),
                ("n_in", C.c_int), ("n_out", C.c_int), ("idx", C.c_int), ("order", C.c_int), ("last_integrator", C.c_int)]


class FllPllFilter(C.Structure):
    """oracle_fll_pll_filter"""
    _fields_ = [(k, C.c_float) for k in ("w", "x", "w0p", "w0p2", "w0p3", "w0f", "w0f2", "a2", "a3", "b3")] + [("order", C.c_int)]


class TrkConf(C.Structure):
    """oracle_trk_conf (same layout as gsh_trk_conf)"""
    _fields_ = [("fs_in", C.c_double), ("code_chip_rate", C.c_double), ("signal_carrier_freq", C.c_double), ("cfo_frequency_hz", C.c_double),
                ("code_length_chips", C.c_uint32), ("code_samples_per_chip", C.c_uint32), ("vector_length", C.c_uint32),
                ("veml", C.c_int32), ("track_pilot", C.c_int32),
                ("early_late_space_chips", C.c_float), ("very_early_late_space_chips", C.c_float),
                ("pll_bw_hz", C.c_float), ("dll_bw_hz", C.c_float), ("fll_bw_hz", C.c_float),
                ("pll_filter_order", C.c_int32), ("dll_filter_order", C.c_int32),
                ("enable_fll_pull_in", C.c_int32), ("enable_fll_steady_state", C.c_int32), ("carrier_aiding", C.c_int32), ("cloop", C.c_int32),
                ("pull_in_time_s", C.c_uint32), ("spc", C.c_float), ("slope", C.c_float), ("y_intercept", C.c_float),
                ("enable_lock_detectors", C.c_int32), ("cn0_samples", C.c_int32), ("cn0_min", C.c_int32), ("max_code_lock_fail", C.c_int32),
                ("max_carrier_lock_fail", C.c_int32), ("cn0_smoother_samples", C.c_int32), ("carrier_lock_test_smoother_samples", C.c_int32),
                ("cn0_smoother_alpha", C.c_float), ("carrier_lock_test_smoother_alpha", C.c_float), ("carrier_lock_th", C.c_double),
                ("enable_symbol_sync", C.c_int32), ("symbols_per_bit", C.c_int32), ("has_secondary", C.c_int32), ("secondary_code_length", C.c_int32),
                ("data_secondary_code_length", C.c_int32), ("extend_correlation_symbols", C.c_int32), ("secondary_code", C.c_uint8 * 320), ("data_secondary_code", C.c_uint8 * 320),
                ("pll_bw_narrow_hz", C.c_float), ("dll_bw_narrow_hz", C.c_float), ("early_late_space_narrow_chips", C.c_float), ("very_early_late_space_narrow_chips", C.c_float),
                ("use_histogram_bit_sync", C.c_int32), ("bs_min_events_for_lock", C.c_int32), ("bs_stable_best_required", C.c_int32),
                ("bs_use_phase_dot_detector", C.c_int32), ("bs_min_prompt_mag", C.c_float), ("enable_bit_sync_time_limit", C.c_int32), ("bs_dominance_ratio", C.c_double),
                ("high_dyn", C.c_int32), ("smoother_length", C.c_uint32), ("bit_synchronization_time_limit_s", C.c_uint32), ("enable_doppler_correction", C.c_int32)]


class TrkEpoch(C.Structure):
    """oracle_trk_epoch (same layout as gsh_trk_epoch)"""
    _fields_ = [("sample_counter", C.c_uint64), ("prn_length_samples", C.c_int32), ("flags", C.c_int32),
                ("corr", C.c_float * 10), ("prompt_data", C.c_float * 2), ("rem_carr_phase_rad", C.c_float), ("cn0_db_hz", C.c_float),
                ("carrier_doppler_hz", C.c_double), ("code_freq_chips", C.c_double), ("carr_phase_error_hz", C.c_double),
                ("carr_freq_error_hz", C.c_double), ("carr_error_filt_hz", C.c_double), ("code_error_chips", C.c_double),
                ("code_error_filt_chips", C.c_double), ("rem_code_phase_samples", C.c_double), ("acc_carrier_phase_rad", C.c_double), ("carrier_lock_test", C.c_double),
                ("state", C.c_int32), ("symbol_flags", C.c_int32), ("p_data_accu", C.c_float * 2),
                ("carrier_phase_rate_step_rad", C.c_double), ("code_phase_rate_step_chips", C.c_double), ("accu", C.c_float * 10)]


_lib = None
_ref = None
_ref_tried = False


def build(quiet: bool = True) -> None:
    """(Re)build liboracle.so and, when /root/reference is present, _ref/."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_gps_l1_ca_code_gen_float.argtypes = [_f32p, C.c_int, C.c_uint]
        L.oracle_gps_l1_ca_code_gen_float.restype = C.c_int
        L.oracle_gps_l1_ca_code_gen_complex_sampled.argtypes = [_f32p, C.c_uint, C.c_int, C.c_uint]
        L.oracle_gps_l1_ca_code_gen_complex_sampled.restype = C.c_int
        L.oracle_code_indices.argtypes = [_i32p, C.c_float, C.c_float, C.c_float, _f32p, C.c_uint, C.c_int, C.c_uint, C.c_int]
        L.oracle_code_indices.restype = None
        L.oracle_mcorr.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int] + [C.c_float] * 6 + [C.c_int, _f32p]
        L.oracle_mcorr.restype = C.c_int
        L.oracle_mcorr_f64.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int] + [C.c_float] * 6 + [C.c_int, _f64p, C.POINTER(C.c_double)]
        L.oracle_mcorr_f64.restype = C.c_int
        L.oracle_sincos.argtypes = [_f32p, C.c_float, C.POINTER(C.c_float), C.c_uint]
        L.oracle_sincos.restype = None
        L.oracle_index_max.argtypes = [_u32p, _f32p, C.c_uint32]
        L.oracle_index_max.restype = None
        L.oracle_mcorr_time.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
        L.oracle_mcorr_time.restype = C.c_double
        # ---- the 16-bit family
        L.oracle_mcorr16_phasors.argtypes = [C.c_float, C.c_float, _f32p]
        L.oracle_mcorr16_phasors.restype = None
        L.oracle_mcorr16_phasor.argtypes = [_i16p, C.c_int, _f32p, C.c_int, _i16p, C.c_int] + [C.c_float] * 6 + [_i16p]
        L.oracle_mcorr16_phasor.restype = C.c_int
        L.oracle_mcorr16.argtypes = [_i16p, C.c_int, _f32p, C.c_int, _i16p, C.c_int] + [C.c_float] * 4 + [_i16p]
        L.oracle_mcorr16.restype = C.c_int
        # ---- loop closure (gnss_oracle_loop.c)
        L.oracle_fll_diff_atan.argtypes = [C.c_float] * 4 + [C.c_double] * 2
        L.oracle_fll_diff_atan.restype = C.c_double
        for name, n in (("oracle_pll_four_quadrant_atan", 2), ("oracle_pll_cloop_two_quadrant_atan", 2),
                        ("oracle_dll_nc_e_minus_l_normalized", 7), ("oracle_dll_nc_vemlp_normalized", 8)):
            getattr(L, name).argtypes = [C.c_float] * n
            getattr(L, name).restype = C.c_double
        L.oracle_loop_filter_design.argtypes = [C.POINTER(LoopFilter), C.c_float, C.c_float, C.c_int, C.c_int]
        L.oracle_loop_filter_initialize.argtypes = [C.POINTER(LoopFilter), C.c_float]
        L.oracle_loop_filter_apply.argtypes = [C.POINTER(LoopFilter), C.c_float]
        L.oracle_loop_filter_apply.restype = C.c_float
        L.oracle_fll_pll_design.argtypes = [C.POINTER(FllPllFilter), C.c_float, C.c_float, C.c_int]
        L.oracle_fll_pll_initialize.argtypes = [C.POINTER(FllPllFilter), C.c_float]
        L.oracle_fll_pll_carrier_error.argtypes = [C.POINTER(FllPllFilter), C.c_float, C.c_float, C.c_float]
        L.oracle_fll_pll_carrier_error.restype = C.c_float
        L.oracle_trk_run.argtypes = [C.POINTER(TrkConf), _f32p, C.c_void_p, C.c_int, _f32p, C.c_uint64, C.c_uint64, C.c_uint64,
                                     C.c_double, C.c_int, C.POINTER(TrkEpoch)]
        L.oracle_trk_run.restype = C.c_int
        L.oracle_trk_run_flags.argtypes = [C.POINTER(TrkConf), _f32p, C.c_void_p, C.c_int, _f32p, C.c_uint64, C.c_uint64, C.c_uint64,
                                           C.c_double, C.c_int, C.POINTER(TrkEpoch), C.c_uint]
        L.oracle_trk_run_flags.restype = C.c_int
        L.oracle_pull_in_over.argtypes = [C.POINTER(TrkConf), C.c_uint64, C.c_uint64]
        L.oracle_pull_in_over.restype = C.c_int
        L.oracle_cn0_m2m4_estimator.argtypes = [_f32p, C.c_int, C.c_float]
        L.oracle_cn0_m2m4_estimator.restype = C.c_float
        L.oracle_carrier_lock_detector.argtypes = [_f32p, C.c_int]
        L.oracle_carrier_lock_detector.restype = C.c_float
        L.oracle_smoother_init.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float]
        L.oracle_smoother_init.restype = None
        L.oracle_smoother_smooth.argtypes = [C.c_void_p, C.c_float]
        L.oracle_smoother_smooth.restype = C.c_float
        _lib = L
    return _lib


def ref():
    """The reference-backed library, or None if oracle/_ref was never built (GPU box without a prebuilt copy)."""
    global _ref, _ref_tried
    if _ref is None and not _ref_tried:
        _ref_tried = True
        path = os.path.join(_HERE, "_ref", "libgnsssdr_ref.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference/src"):
            try:
                build()
            except Exception:  # pragma: no cover
                pass
        if os.path.exists(path):
            R = C.CDLL(path)
            R.ref_simd_supported.restype = C.c_int
            R.ref_set_flavour.argtypes = [C.c_int]
            R.ref_mcorr_run.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int] + [C.c_float] * 6 + [C.c_int, _f32p]
            R.ref_mcorr_run.restype = C.c_int
            R.ref_mcorr_time.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
            R.ref_mcorr_time.restype = C.c_double
            R.ref_sincos_generic.argtypes = [_f32p, C.c_float, C.POINTER(C.c_float), C.c_uint]
            R.ref_index_max_generic.argtypes = [_u32p, _f32p, C.c_uint32]
            R.ref_gps_l1_ca_code_gen_float.argtypes = [_f32p, C.c_int, C.c_uint]
            R.ref_gps_l1_ca_code_gen_complex_sampled.argtypes = [_f32p, C.c_int, C.c_uint, C.c_int, C.c_uint]
            R.ref_galileo_e1_code_gen_sinboc11_float.argtypes = [_f32p, C.c_char_p, C.c_uint]
            R.ref_galileo_e1_code_gen_complex_sampled.argtypes = [_f32p, C.c_int, C.c_char_p, C.c_int, C.c_uint, C.c_int, C.c_uint]
            R.ref_gps_l5i_code_gen_float.argtypes = [_f32p, C.c_uint]
            R.ref_gps_l5q_code_gen_float.argtypes = [_f32p, C.c_uint]
            R.ref_resampler_generic.argtypes = [C.POINTER(C.POINTER(C.c_float)), _f32p, C.c_float, C.c_float, _f32p, C.c_uint, C.c_int, C.c_uint]
            R.ref_hd_resampler_generic.argtypes = [C.POINTER(C.POINTER(C.c_float)), _f32p, C.c_float, C.c_float, C.c_float, _f32p, C.c_uint, C.c_int, C.c_uint]
            if hasattr(R, "ref_mcorr16_run"):  # the 16-bit family
                R.ref_mcorr16_run.argtypes = [_i16p, C.c_int, _f32p, C.c_int, _i16p, C.c_int] + [C.c_float] * 4 + [_i16p]
                R.ref_mcorr16_run.restype = C.c_int
                R.ref_mcorr16_phasors.argtypes = [C.c_float, C.c_float, _f32p]
                R.ref_mcorr16_phasors.restype = None
                R.ref_mcorr16_time.argtypes = [_i16p, C.c_int, _f32p, C.c_int, _i16p, C.c_long, C.c_int, C.c_int] + [C.c_float] * 4 + [_i16p]
                R.ref_mcorr16_time.restype = C.c_double
            if hasattr(R, "ref_fll_diff_atan"):  # loop-closure objects (added with SURVEY 8f-1)
                R.ref_fll_diff_atan.argtypes = [C.c_float] * 4 + [C.c_double] * 2
                R.ref_fll_diff_atan.restype = C.c_double
                for name, n in (("ref_pll_four_quadrant_atan", 2), ("ref_pll_cloop_two_quadrant_atan", 2),
                                ("ref_dll_nc_e_minus_l_normalized", 7), ("ref_dll_nc_vemlp_normalized", 8)):
                    getattr(R, name).argtypes = [C.c_float] * n
                    getattr(R, name).restype = C.c_double
                R.ref_loop_filter_run.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, _f32p, _f32p, C.c_int]
                R.ref_fll_pll_filter_run.argtypes = [C.c_float, C.c_float, C.c_int, C.c_float, _f32p, _f32p, C.c_float, _f32p, C.c_int]
            if hasattr(R, "ref_bit_sync_run"):
                _i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
                R.ref_bit_sync_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_int, _f32p, _i32p, C.c_int, _i32p, _i32p, _i32p]
            if hasattr(R, "ref_smoother_run"):  # lock detectors + smoother (added with SURVEY 8f-2)
                R.ref_cn0_m2m4_estimator.argtypes = [_f32p, C.c_int, C.c_float]
                R.ref_cn0_m2m4_estimator.restype = C.c_float
                R.ref_carrier_lock_detector.argtypes = [_f32p, C.c_int]
                R.ref_carrier_lock_detector.restype = C.c_float
                R.ref_smoother_run.argtypes = [C.c_float, C.c_int, C.c_float, C.c_float, _f32p, C.c_int, _f32p]
            _ref = R
    return _ref


# --------------------------------------------------------------------------- numpy-level helpers

def _iq(x: np.ndarray) -> np.ndarray:
    """complex64 array -> contiguous float32 view [.., 2]."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    return x.view(np.float32)


def ca_code(prn: int, chip_shift: int = 0) -> np.ndarray:
    out = np.empty(1023, np.float32)
    if lib().oracle_gps_l1_ca_code_gen_float(out, prn, chip_shift) != 0:
        raise ValueError(f"invalid GPS PRN {prn}")
    return out


def ca_code_complex_sampled(prn: int, fs: int, chip_shift: int = 0) -> np.ndarray:
    n = int(fs / (1023000.0 / 1023.0))
    out = np.zeros(2 * n, np.float32)
    r = lib().oracle_gps_l1_ca_code_gen_complex_sampled(out, prn, fs, chip_shift)
    if r < 0:
        raise ValueError(f"invalid GPS PRN {prn}")
    return out.view(np.complex64)


def code_indices(n, shifts, rem_code, code_step, code_rate_step=0.0, code_len=1023, high_dyn=False) -> np.ndarray:
    shifts = np.ascontiguousarray(shifts, np.float32)
    idx = np.empty((len(shifts), n), np.int32)
    lib().oracle_code_indices(idx, rem_code, code_step, code_rate_step, shifts, code_len, len(shifts), n, int(high_dyn))
    return idx


def mcorr(code, shifts, x, rem_carr, phase_step, rem_code, code_step, phase_rate_step=0.0, code_rate_step=0.0,
          high_dyn=False) -> np.ndarray:
    """float32 oracle (reference _generic order).  Returns complex64[n_taps]."""
    code = np.ascontiguousarray(code, np.float32)
    shifts = np.ascontiguousarray(shifts, np.float32)
    xi = _iq(x)
    out = np.empty(2 * len(shifts), np.float32)
    rc = lib().oracle_mcorr(code, len(code), shifts, len(shifts), xi, len(xi) // 2, rem_carr, phase_step,
                            phase_rate_step, rem_code, code_step, code_rate_step, int(high_dyn), out)
    if rc != 0:
        raise RuntimeError(f"oracle_mcorr failed: {rc}")
    return out.view(np.complex64)


def mcorr_f64(code, shifts, x, rem_carr, phase_step, rem_code, code_step, phase_rate_step=0.0, code_rate_step=0.0,
              high_dyn=False):
    """float64 truth.  Returns (complex128[n_taps], sum_abs)."""
    code = np.ascontiguousarray(code, np.float32)
    shifts = np.ascontiguousarray(shifts, np.float32)
    xi = _iq(x)
    out = np.empty(2 * len(shifts), np.float64)
    sabs = C.c_double(0.0)
    rc = lib().oracle_mcorr_f64(code, len(code), shifts, len(shifts), xi, len(xi) // 2, rem_carr, phase_step,
                                phase_rate_step, rem_code, code_step, code_rate_step, int(high_dyn), out, C.byref(sabs))
    if rc != 0:
        raise RuntimeError(f"oracle_mcorr_f64 failed: {rc}")
    return out.view(np.complex128), sabs.value


def ref_mcorr(code, shifts, x, rem_carr, phase_step, rem_code, code_step, phase_rate_step=0.0, code_rate_step=0.0,
              high_dyn=False, simd=False) -> np.ndarray:
    """The reference's own Cpu_Multicorrelator_Real_Codes (oracle/_ref)."""
    R = ref()
    if R is None:
        raise RuntimeError("oracle/_ref is not built")
    code = np.ascontiguousarray(code, np.float32)
    shifts = np.ascontiguousarray(shifts, np.float32)
    xi = _iq(x)
    out = np.empty(2 * len(shifts), np.float32)
    R.ref_set_flavour(int(simd))
    try:
        R.ref_mcorr_run(code, len(code), shifts, len(shifts), xi, len(xi) // 2, rem_carr, phase_step,
                        phase_rate_step, rem_code, code_step, code_rate_step, int(high_dyn), out)
    finally:
        R.ref_set_flavour(0)
    return out.view(np.complex64)


# --------------------------------------------------------------------------- the 16-bit family

def _iq16(x) -> np.ndarray:
    """int16 I/Q as a contiguous [n, 2] array (accepts [n, 2] integers or a complex array with integer parts)."""
    x = np.asarray(x)
    if np.iscomplexobj(x):
        x = np.stack([x.real, x.imag], axis=-1)
    return np.ascontiguousarray(x, dtype=np.int16).reshape(-1, 2)


def mcorr16_phasors(rem_carr: float, phase_step: float) -> np.ndarray:
    """(phase0 re, im, increment re, im) as Cpu_Multicorrelator_16sc forms them before the kernel call."""
    out = np.empty(4, np.float32)
    lib().oracle_mcorr16_phasors(rem_carr, phase_step, out)
    return out


def mcorr16(code_iq, shifts, x_iq, rem_carr, phase_step, rem_code, code_step) -> np.ndarray:
    """One Cpu_Multicorrelator_16sc call, restated (generic protokernels).  Returns int16[n_taps, 2]."""
    code = _iq16(code_iq)
    xi = _iq16(x_iq)
    shifts = np.ascontiguousarray(shifts, np.float32)
    out = np.empty((len(shifts), 2), np.int16)
    rc = lib().oracle_mcorr16(code.reshape(-1), len(code), shifts, len(shifts), xi.reshape(-1), len(xi), rem_carr, phase_step, rem_code, code_step, out.reshape(-1))
    if rc != 0:
        raise RuntimeError(f"oracle_mcorr16 failed: {rc}")
    return out


def ref_mcorr16(code_iq, shifts, x_iq, rem_carr, phase_step, rem_code, code_step, simd=False) -> np.ndarray:
    """The reference's own Cpu_Multicorrelator_16sc (oracle/_ref)."""
    R = ref()
    if R is None or not hasattr(R, "ref_mcorr16_run"):
        raise RuntimeError("oracle/_ref is not built (or predates the 16-bit family)")
    code = _iq16(code_iq)
    xi = _iq16(x_iq)
    shifts = np.ascontiguousarray(shifts, np.float32)
    out = np.empty((len(shifts), 2), np.int16)
    R.ref_set_flavour(int(simd))
    try:
        R.ref_mcorr16_run(code.reshape(-1), len(code), shifts, len(shifts), xi.reshape(-1), len(xi), rem_carr, phase_step, rem_code, code_step, out.reshape(-1))
    finally:
        R.ref_set_flavour(0)
    return out


# --------------------------------------------------------------------------- loop closure helpers

def loop_filter_run(update_interval, noise_bandwidth, order, include_last_integrator, initial_output, x):
    """Tracking_loop_filter: design, initialize(initial_output), apply every element of x"""
    f = LoopFilter()
    L = lib()
    L.oracle_loop_filter_design(C.byref(f), update_interval, noise_bandwidth, order, int(include_last_integrator))
    L.oracle_loop_filter_initialize(C.byref(f), initial_output)
    return np.array([L.oracle_loop_filter_apply(C.byref(f), float(v)) for v in np.asarray(x, np.float32)], np.float32)


def fll_pll_filter_run(fll_bw_hz, pll_bw_hz, order, acq_doppler_hz, fll_disc, pll_disc, correlation_time_s):
    f = FllPllFilter()
    L = lib()
    L.oracle_fll_pll_design(C.byref(f), fll_bw_hz, pll_bw_hz, order)
    L.oracle_fll_pll_initialize(C.byref(f), acq_doppler_hz)
    return np.array([L.oracle_fll_pll_carrier_error(C.byref(f), float(a), float(b), correlation_time_s)
                     for a, b in zip(np.asarray(fll_disc, np.float32), np.asarray(pll_disc, np.float32))], np.float32)


class Resampler(C.Structure):
    """oracle_resampler"""
    _fields_ = [("fs_in", C.c_double), ("fs_out", C.c_double), ("phase", C.c_uint32), ("lphase", C.c_uint32), ("phase_step", C.c_uint32)]


def direct_resampler(x: np.ndarray, fs_in: float, fs_out: float, call_sizes=None) -> np.ndarray:
    """direct_resampler_conditioner_cc run over the whole of x as a sequence of general_work calls asking for call_sizes outputs each
    (default: one call).  Returns the concatenated output."""
    L = lib()
    L.oracle_direct_resampler_init.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.oracle_direct_resampler_init.restype = None
    L.oracle_direct_resampler_work.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, C.POINTER(C.c_int)]
    L.oracle_direct_resampler_work.restype = C.c_int
    r = Resampler()
    L.oracle_direct_resampler_init(C.byref(r), fs_in, fs_out)
    xi = _iq(x).reshape(-1)
    cap = int(len(x) * max(1.0, fs_out / fs_in)) + 8
    sizes = list(call_sizes) if call_sizes is not None else [cap]
    outs, pos, k = [], 0, 0
    while pos < len(x):
        want = sizes[k % len(sizes)]
        k += 1
        buf = np.zeros(2 * want, np.float32)
        cons = C.c_int(0)
        rest = np.ascontiguousarray(xi[2 * pos:])
        n = L.oracle_direct_resampler_work(C.byref(r), rest, len(x) - pos, buf, want, C.byref(cons))
        outs.append(buf[:2 * n].copy())
        if n == 0 and cons.value == 0:
            break
        pos += cons.value
        if fs_in < fs_out and n < want:
            break
    out = np.concatenate(outs) if outs else np.zeros(0, np.float32)
    return out.view(np.complex64)


def bit_sync_run(prompts: np.ndarray, bins: int, min_events_for_lock=10, stable_best_required=3, dominance_ratio=0.6, min_prompt_mag=0.0,
                 use_phase_dot_detector=True, quality_ok=None):
    """HistogramBitSynchronizer (T/bit_synchronizer.cc) restated in C: -> (lock_event[n], edge_phase[n], epochs_until_next_edge[n])"""
    L = lib()
    L.oracle_bit_sync_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_int]
    L.oracle_bit_sync_init.restype = None
    L.oracle_bit_sync_update.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
    L.oracle_bit_sync_epochs_until_next_edge.argtypes = [C.c_void_p]
    buf = C.create_string_buffer(1024)
    L.oracle_bit_sync_init(buf, bins, min_events_for_lock, stable_best_required, dominance_ratio, min_prompt_mag, int(use_phase_dot_detector))
    p = np.ascontiguousarray(prompts, np.complex64)
    q = np.ones(len(p), np.int32) if quality_ok is None else np.asarray(quality_ok, np.int32)
    ev, un = np.zeros(len(p), np.int32), np.zeros(len(p), np.int32)
    for i, v in enumerate(p):
        ev[i] = L.oracle_bit_sync_update(buf, float(v.real), float(v.imag), int(q[i]))
        un[i] = L.oracle_bit_sync_epochs_until_next_edge(buf)
    return ev, un


class Smoother(C.Structure):
    """oracle_smoother"""
    _fields_ = [(k, C.c_float) for k in ("alpha", "one_minus_alpha", "old_value", "min_value", "offset", "init_sum")] + [
        (k, C.c_int) for k in ("samples_for_initialization", "init_counter", "initializing")]


def cn0_m2m4_estimator(prompt: np.ndarray, coh_integration_time_s: float) -> float:
    p = _iq(prompt)
    return float(lib().oracle_cn0_m2m4_estimator(p.reshape(-1), len(prompt), coh_integration_time_s))


def carrier_lock_detector(prompt: np.ndarray, length: int | None = None) -> float:
    p = _iq(prompt)
    return float(lib().oracle_carrier_lock_detector(p.reshape(-1), len(prompt) if length is None else length))


def smoother_run(alpha: float, samples_for_initialization: int, raw, min_value: float = 25.0, offset: float = 12.0) -> np.ndarray:
    """Exponential_Smoother configured as trk.cc:680-692 does, fed with `raw`"""
    L = lib()
    s = Smoother()
    L.oracle_smoother_init(C.byref(s), alpha, samples_for_initialization, min_value, offset)
    return np.array([L.oracle_smoother_smooth(C.byref(s), float(v)) for v in np.asarray(raw, np.float32)], np.float32)


def trk_conf(**kw) -> TrkConf:
    """oracle_trk_conf with Dll_Pll_Conf's defaults (dll_pll_conf.h:33-90) for the fields that have one"""
    c = TrkConf()
    d = dict(fs_in=4e6, code_chip_rate=1.023e6, signal_carrier_freq=1575.42e6, cfo_frequency_hz=0.0, code_length_chips=1023,
             code_samples_per_chip=1, vector_length=4000, veml=0, track_pilot=0, early_late_space_chips=0.5,
             very_early_late_space_chips=0.6, pll_bw_hz=35.0, dll_bw_hz=2.0, fll_bw_hz=35.0, pll_filter_order=3, dll_filter_order=2,
             enable_fll_pull_in=0, enable_fll_steady_state=0, carrier_aiding=1, cloop=1, pull_in_time_s=5, spc=0.5, slope=1.0,
             y_intercept=1.0,
             # lock detectors / C/N0: Dll_Pll_Conf defaults (gnss_sdr_flags.cc:44-53, dll_pll_conf.h:58-59,70-71); off unless asked for
             enable_lock_detectors=0, cn0_samples=20, cn0_min=25, max_code_lock_fail=50, max_carrier_lock_fail=5000,
             cn0_smoother_samples=200, carrier_lock_test_smoother_samples=25, cn0_smoother_alpha=0.002,
             carrier_lock_test_smoother_alpha=0.002, carrier_lock_th=0.7,
             enable_symbol_sync=0, symbols_per_bit=0, has_secondary=0, secondary_code_length=0, data_secondary_code_length=0,
             # extended integration: Dll_Pll_Conf defaults (dll_pll_conf.h:49-54, 68)
             extend_correlation_symbols=1, pll_bw_narrow_hz=5.0, dll_bw_narrow_hz=0.75, early_late_space_narrow_chips=0.15,
             very_early_late_space_narrow_chips=0.5,
             # histogram bit synchroniser: Dll_Pll_Conf defaults (dll_pll_conf.h:43,60,75-76,88); the block switches it on for signals
             # without a secondary code and more than one symbol per bit (trk.cc:1389) -- here the caller does
             use_histogram_bit_sync=0, bs_min_events_for_lock=10, bs_stable_best_required=3, bs_use_phase_dot_detector=1,
             bs_min_prompt_mag=0.0, bs_dominance_ratio=0.6,
             high_dyn=0, smoother_length=10, enable_bit_sync_time_limit=0, bit_synchronization_time_limit_s=20, enable_doppler_correction=0)
    d.update(kw)
    for k, v in d.items():
        setattr(c, k, v)
    return c


def trk_run(conf: TrkConf, code, x, start_sample, acq_sample_stamp, acq_doppler_hz, n_epochs, data_code=None, pull_in_over=False):
    """closed DLL/PLL loop of one channel on the CPU; returns the list of completed TrkEpoch records.  pull_in_over: the pull-in transitory was over at the
    pull-in call already (a read pointer behind the acquisition's stamp wraps trk.cc:1912's unsigned difference)"""
    code = np.ascontiguousarray(code, np.float32)
    rec = (TrkEpoch * n_epochs)()
    dc = None
    if data_code is not None:
        data_code = np.ascontiguousarray(data_code, np.float32)
        dc = data_code.ctypes.data_as(C.c_void_p)
    n = lib().oracle_trk_run_flags(C.byref(conf), code, dc, len(code), _iq(x), len(x), int(start_sample), int(acq_sample_stamp),
                                   float(acq_doppler_hz), n_epochs, rec, 1 if pull_in_over else 0)
    return list(rec)[:n]


def set_symbol_sync(conf, symbols_per_bit: int, secondary_code: str = "", has_secondary: bool = False, data_secondary_code: str = "") -> None:
    """Fill the symbol-synchronisation fields as the tracking block's constructor does per signal (trk.cc:196-300): e.g. GPS L1 C/A:
    symbols_per_bit = 20, secondary_code = the 160-symbol telemetry preamble, has_secondary = False; Galileo E1 pilot: symbols_per_bit = 1,
    secondary_code = the 25-chip E1C code, has_secondary = True."""
    conf.enable_symbol_sync = 1
    conf.symbols_per_bit = symbols_per_bit
    conf.has_secondary = int(has_secondary)
    conf.secondary_code_length = len(secondary_code)
    conf.data_secondary_code_length = len(data_secondary_code)
    for i, ch in enumerate(secondary_code.encode()):
        conf.secondary_code[i] = ch
    for i, ch in enumerate(data_secondary_code.encode()):
        conf.data_secondary_code[i] = ch
