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
        L.refconv_create.restype = C.c_void_p
        L.refconv_create.argtypes = [C.c_int, C.c_double, C.c_double]
        L.refconv_destroy.argtypes = [C.c_void_p]
        L.refconv_item_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.refconv_forecast.restype = C.c_int
        L.refconv_forecast.argtypes = [C.c_void_p, C.c_int]
        L.refconv_general_work.restype = C.c_int
        L.refconv_general_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
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


# ---- the reference's front-end blocks (round 6): direct resamplers and data-type adapters, driven through general_work
K_RESAMPLER_CC, K_RESAMPLER_CB, K_RESAMPLER_CS, K_CSHORT_TO_GR_COMPLEX, K_IBYTE_TO_CBYTE, K_IBYTE_TO_CSHORT, K_ISHORT_TO_CSHORT = range(3, 10)
_CONV_DTYPES = {K_RESAMPLER_CC: (np.complex64, 1, np.complex64, 1), K_RESAMPLER_CB: (np.int8, 2, np.int8, 2), K_RESAMPLER_CS: (np.int16, 2, np.int16, 2),
                K_CSHORT_TO_GR_COMPLEX: (np.int16, 2, np.complex64, 1), K_IBYTE_TO_CBYTE: (np.int8, 1, np.int8, 2), K_IBYTE_TO_CSHORT: (np.int8, 1, np.int16, 2),
                K_ISHORT_TO_CSHORT: (np.int16, 1, np.int16, 2)}


class RefConvBlock:
    """One of the reference's resampler / data-type adapter blocks.  Items are handed over as numpy arrays of the block's item type: complex64 for gr_complex,
    int8 / int16 arrays of shape (n, 2) for lv_8sc_t / lv_16sc_t, flat int8 / int16 for the interleaved inputs."""

    def __init__(self, kind: int, fs_in: float = 0.0, fs_out: float = 0.0):
        self.h = lib().refconv_create(kind, fs_in, fs_out)
        if not self.h:
            raise RuntimeError("reference block construction failed")
        self.kind = kind
        self.in_dtype, self.in_words, self.out_dtype, self.out_words = _CONV_DTYPES[kind]
        a, b = C.c_int(0), C.c_int(0)
        lib().refconv_item_sizes(self.h, C.byref(a), C.byref(b))
        assert a.value == np.dtype(self.in_dtype).itemsize * self.in_words and b.value == np.dtype(self.out_dtype).itemsize * self.out_words

    def close(self):
        if self.h:
            lib().refconv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forecast(self, noutput_items: int) -> int:
        return lib().refconv_forecast(self.h, noutput_items)

    def work(self, x: np.ndarray, noutput_items: int):
        """One general_work call: (outputs, items consumed)."""
        x = np.ascontiguousarray(x, self.in_dtype)
        n_items = x.size // self.in_words
        out = np.zeros((max(noutput_items, 1), self.out_words) if self.out_words > 1 else max(noutput_items, 1), self.out_dtype)
        consumed = C.c_int(0)
        r = lib().refconv_general_work(self.h, x.ctypes.data, n_items, noutput_items, out.ctypes.data, C.byref(consumed))
        return out[:r].copy(), consumed.value

    def run(self, x: np.ndarray, call_sizes=(4096,), offer: str = "all"):
        """The whole of x as the scheduler would feed it: calls asking for call_sizes outputs in turn (cycled); unconsumed items stay in front of the next call.
        offer = "all": every call sees everything that is left (a scheduler with a full buffer); "forecast": exactly what the block's forecast asks for, the
        least a scheduler may offer.  Stops when a call could not be given its forecast."""
        x = np.ascontiguousarray(x, self.in_dtype)
        n_items = x.size // self.in_words
        flat = x.reshape(n_items, self.in_words) if self.in_words > 1 else x
        outs, pos, k = [], 0, 0
        while True:
            want = int(call_sizes[k % len(call_sizes)])
            k += 1
            need = self.forecast(want)
            if pos + need > n_items:
                break
            y, cons = self.work(flat[pos:] if offer == "all" else flat[pos:pos + need], want)
            outs.append(y)
            pos += cons
        if not outs:
            return np.zeros((0, self.out_words) if self.out_words > 1 else 0, self.out_dtype), 0
        return np.concatenate(outs, axis=0), pos
