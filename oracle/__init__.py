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
