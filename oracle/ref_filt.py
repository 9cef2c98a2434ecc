"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/_ref/libgnsssdr_ref_filt.so: the reference's own pulse_blanking_cc, Notch and NotchLite blocks
(src/algorithms/input_filter/gnuradio_blocks/) compiled in place by oracle/Makefile against the GNU Radio mock, driven through general_work."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

K_PULSE_BLANKING, K_NOTCH, K_NOTCH_LITE = range(3)
_LIB = None


def available() -> bool:
    return os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libgnsssdr_ref_filt.so"))


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libgnsssdr_ref_filt.so"))
        L.reffilt_create.restype = C.c_void_p
        L.reffilt_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.reffilt_destroy.argtypes = [C.c_void_p]
        L.reffilt_general_work.restype = C.c_int
        L.reffilt_general_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.reffilt_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 2 + [C.POINTER(C.c_int32)] * 2 + [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        _LIB = L
    return _LIB


class RefFilterBlock:
    def __init__(self, kind: int, pfa: float, p_c_factor: float = 0.9, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000,
                 n_segments_coeff: int = 0):
        self.h = lib().reffilt_create(kind, pfa, p_c_factor, length, n_segments_est, n_segments_reset, n_segments_coeff)
        if not self.h:
            raise RuntimeError("reference filter block construction failed")
        self.kind = kind

    def close(self):
        if self.h:
            lib().reffilt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def work(self, x: np.ndarray, noutput_items: int | None = None):
        """One general_work call over the items x (history, if the block has any, already in front).  Returns (outputs, consumed)."""
        x = np.ascontiguousarray(x, np.complex64)
        nout = len(x) if noutput_items is None else noutput_items
        out = np.zeros(max(nout, 1), np.complex64)
        consumed = C.c_int(0)
        r = lib().reffilt_general_work(self.h, x.ctypes.data, len(x), nout, out.ctypes.data, C.byref(consumed))
        return out[:r].copy(), consumed.value

    def state(self) -> dict:
        thres, npw = C.c_float(0), C.c_float(0)
        nseg, fst, ncoef = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        lo = (C.c_float * 2)()
        z0 = (C.c_float * 2)()
        lib().reffilt_state(self.h, C.byref(thres), C.byref(npw), C.byref(nseg), C.byref(fst), lo, C.byref(ncoef), z0)
        return dict(thres=thres.value, noise_pow_est=npw.value, n_segments=nseg.value, filter_state=bool(fst.value), last_out=complex(lo[0], lo[1]),
                    n_segments_coeff=ncoef.value, z0=complex(z0[0], z0[1]))
