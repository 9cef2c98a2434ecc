"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/_ref/libgnsssdr_ref_acq.so: the REFERENCE's own acquisition blocks
(pcps_acquisition.cc and the detector blocks, compiled from /root/reference by oracle/Makefile against stand-ins for GNU Radio /
VOLK / FFTW, see oracle/ref_acq_api.cc) driven through ``general_work`` like the GNU Radio scheduler does.

Used to PIN oracle/pcps_oracle.py (tests/test_pcps_oracle_pinned.py) and as a second, independent checker of the HIP path.
The product never imports this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libgnsssdr_ref_acq.so")

K_PCPS, K_TONG, K_8MS, K_CCCWSR, K_QUICKSYNC, K_FINE_DOPPLER, K_E5A_CAF = range(7)


class Status(C.Structure):
    _fields_ = [("state", C.c_int32), ("active", C.c_int32), ("step_two", C.c_int32), ("positive_acq", C.c_int32),
                ("dwell_count", C.c_uint32), ("tong_count", C.c_uint32), ("num_doppler_bins", C.c_uint32), ("fft_size", C.c_uint32),
                ("effective_fft_size", C.c_uint32), ("consumed_samples", C.c_uint32), ("code_phase", C.c_uint32), ("doppler_bins_step2", C.c_uint32),
                ("sample_counter", C.c_uint64),
                ("mag", C.c_float), ("input_power", C.c_float), ("test_statistics", C.c_float), ("threshold", C.c_float),
                ("threshold_step_two", C.c_float), ("doppler_center_step_two", C.c_float),
                ("acq_delay_samples", C.c_double), ("acq_doppler_hz", C.c_double), ("acq_samplestamp_samples", C.c_uint64),
                ("acq_doppler_step", C.c_uint32), ("fs", C.c_int64),
                ("conf_fs_in", C.c_int64), ("conf_resampled_fs", C.c_int64),
                ("conf_samples_per_ms", C.c_float), ("conf_samples_per_code", C.c_float), ("conf_resampler_ratio", C.c_float),
                ("conf_threshold", C.c_float), ("conf_pfa", C.c_float), ("conf_pfa2", C.c_float), ("conf_doppler_step2", C.c_float),
                ("conf_samples_per_chip", C.c_uint32), ("conf_doppler_max", C.c_uint32), ("conf_doppler_step", C.c_uint32),
                ("conf_sampled_ms", C.c_uint32), ("conf_ms_per_code", C.c_uint32), ("conf_max_dwells", C.c_uint32),
                ("conf_num_doppler_bins_step2", C.c_uint32),
                ("conf_it_size", C.c_int32), ("conf_use_cfar", C.c_int32), ("conf_bit_transition_flag", C.c_int32),
                ("conf_make_2_steps", C.c_int32), ("conf_blocking", C.c_int32), ("conf_use_automatic_resampler", C.c_int32),
                ("consumed_last", C.c_int32), ("consumed_total", C.c_int64), ("n_events", C.c_int32), ("events", C.c_int32 * 32)]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "events"}
        d["events"] = list(self.events[:self.n_events])
        return d


class Override(C.Structure):
    _names = ("samples_per_ms", "samples_per_code", "samples_per_chip", "sampled_ms", "threshold", "doppler_step", "doppler_max", "max_dwells",
              "bit_transition_flag", "dump", "code_length", "vector_length", "num_codes")
    _fields_ = [("has_" + n, C.c_int32) for n in _names] + [
        ("samples_per_ms", C.c_float), ("samples_per_code", C.c_float), ("threshold", C.c_float),
        ("samples_per_chip", C.c_uint32), ("sampled_ms", C.c_uint32), ("doppler_step", C.c_uint32), ("doppler_max", C.c_uint32),
        ("max_dwells", C.c_uint32), ("bit_transition_flag", C.c_int32), ("dump", C.c_int32),
        ("code_length", C.c_uint32), ("vector_length", C.c_uint32), ("num_codes", C.c_uint32)]


FFT_HOOK = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int)

_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.refacq_create.restype = C.c_void_p
        L.refacq_create.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_double, C.c_double,
                                    C.c_uint32, C.POINTER(C.c_int32)]
        L.refacq_create_with_override.restype = C.c_void_p
        L.refacq_create_with_override.argtypes = L.refacq_create.argtypes + [C.POINTER(Override)]
        L.refacq_destroy.argtypes = [C.c_void_p]
        L.refacq_set_satellite.argtypes = [C.c_void_p, C.c_char, C.c_char_p, C.c_uint32]
        L.refacq_set_channel.argtypes = [C.c_void_p, C.c_uint32]
        L.refacq_set_local_code.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refacq_set_active.argtypes = [C.c_void_p, C.c_int]
        L.refacq_set_doppler_center.argtypes = [C.c_void_p, C.c_int32]
        L.refacq_set_resampler_latency.argtypes = [C.c_void_p, C.c_uint32]
        L.refacq_general_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.refacq_forecast.argtypes = [C.c_void_p, C.c_int]
        L.refacq_get_status.argtypes = [C.c_void_p, C.POINTER(Status)]
        L.refacq_clear_events.argtypes = [C.c_void_p]
        L.refacq_read_grid.restype = C.c_int64
        L.refacq_read_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.refacq_read_fft_codes.restype = C.c_int64
        L.refacq_read_fft_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.refacq_read_wipeoff.restype = C.c_int64
        L.refacq_read_wipeoff.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int64]
        L.ref_fft_set_hook.argtypes = [C.c_void_p]
        L.ref_fft_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


_hook_keepalive = None


def set_fft(kind: str = "double") -> None:
    """Choose the transform behind the blocks' gr::fft objects: "double" = ref_fft.cc (exact-definition DFT in float64, rounded once);
    "pocketfft32" = scipy's single-precision pocketfft, the transform oracle/pcps_oracle.py uses -- with it, block and restatement must
    agree value for value."""
    global _hook_keepalive
    L = lib()
    if kind == "double":
        L.ref_fft_set_hook(None)
        _hook_keepalive = None
        return
    import scipy.fft

    def hook(pin, pout, n, forward):
        a = np.ctypeslib.as_array(pin, shape=(2 * n,)).view(np.complex64)
        out = np.ctypeslib.as_array(pout, shape=(2 * n,)).view(np.complex64)
        if forward:
            out[:] = scipy.fft.fft(a)
        else:
            out[:] = scipy.fft.ifft(a, norm="forward")

    _hook_keepalive = FFT_HOOK(hook)
    L.ref_fft_set_hook(C.cast(_hook_keepalive, C.c_void_p))


def fft(x: np.ndarray, forward: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty_like(x)
    lib().ref_fft_execute(x.ctypes.data, out.ctypes.data, len(x), 1 if forward else 0)
    return out


class RefAcqBlock:
    """One reference acquisition block + its Gnss_Synchro, configured from properties as an adapter would
    (base_pcps_acquisition.cc:40-49: ms_per_code, sampled_ms default, Acq_Conf::SetFromConfiguration)."""

    def __init__(self, kind: int, props: dict, chip_rate: float, opt_freq: float, ms_per_code: int, role: str = "Acquisition",
                 extra=(0, 0, 0), override: dict | None = None, system: str = "G", signal: str = "1C", prn: int = 1):
        L = lib()
        keys = (C.c_char_p * len(props))(*[k.encode() for k in props])
        vals = (C.c_char_p * len(props))(*[str(v).encode() for v in props.values()])
        ex = (C.c_int32 * 3)(*[int(e) for e in (list(extra) + [0, 0, 0])[:3]])
        if override:
            ov = Override()
            for k, v in override.items():
                setattr(ov, "has_" + k, 1)
                setattr(ov, k, v)
            self.h = L.refacq_create_with_override(kind, role.encode(), keys, vals, len(props), chip_rate, opt_freq, ms_per_code, ex, C.byref(ov))
        else:
            self.h = L.refacq_create(kind, role.encode(), keys, vals, len(props), chip_rate, opt_freq, ms_per_code, ex)
        if not self.h:
            raise RuntimeError("reference block construction failed")
        self.kind = kind
        L.refacq_set_satellite(self.h, system.encode(), signal.encode(), prn)
        self._codes = None

    def close(self):
        if self.h:
            lib().refacq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_satellite(self, system: str, signal: str, prn: int):
        lib().refacq_set_satellite(self.h, system.encode(), signal.encode(), prn)

    def set_local_code(self, code: np.ndarray, code2: np.ndarray | None = None):
        c1 = np.ascontiguousarray(code, np.complex64)
        c2 = None if code2 is None else np.ascontiguousarray(code2, np.complex64)
        self._codes = (c1, c2)
        lib().refacq_set_local_code(self.h, c1.ctypes.data, None if c2 is None else c2.ctypes.data)

    def set_active(self, active: bool = True):
        lib().refacq_set_active(self.h, 1 if active else 0)

    def set_doppler_center(self, center: int):
        assert lib().refacq_set_doppler_center(self.h, center) == 0

    def set_resampler_latency(self, n: int):
        assert lib().refacq_set_resampler_latency(self.h, n) == 0

    def set_channel(self, ch: int):
        lib().refacq_set_channel(self.h, ch)

    def work(self, items: np.ndarray, noutput_items: int = 1) -> tuple[int, int]:
        """One general_work call with len(items) available input items; returns (return value, consumed)."""
        items = np.ascontiguousarray(items)
        consumed = C.c_int(0)
        r = lib().refacq_general_work(self.h, items.ctypes.data, len(items), noutput_items, C.byref(consumed))
        return r, consumed.value

    def forecast(self, noutput: int = 1) -> int:
        return lib().refacq_forecast(self.h, noutput)

    def status(self) -> dict:
        st = Status()
        lib().refacq_get_status(self.h, C.byref(st))
        return st.as_dict()

    def clear_events(self):
        lib().refacq_clear_events(self.h)

    def grid(self) -> np.ndarray:
        st = self.status()
        if self.kind == K_PCPS:
            bins = st["doppler_bins_step2"] if st["step_two"] else st["num_doppler_bins"]
            width = st["effective_fft_size"]
        else:
            bins, width = st["num_doppler_bins"], st["fft_size"]
        out = np.empty(bins * width, np.float32)
        n = lib().refacq_read_grid(self.h, out.ctypes.data, out.size)
        return out[:n].reshape(-1, width)

    def fft_codes(self) -> np.ndarray:
        st = self.status()
        out = np.empty(st["fft_size"], np.complex64)
        n = lib().refacq_read_fft_codes(self.h, out.ctypes.data, out.size)
        return out[:n]

    def wipeoff(self, bin_index: int, step_two: bool = False) -> np.ndarray:
        st = self.status()
        out = np.empty(st["fft_size"], np.complex64)
        n = lib().refacq_read_wipeoff(self.h, bin_index, 1 if step_two else 0, out.ctypes.data, out.size)
        return out[:n]

    def run_stream(self, x: np.ndarray, chunk: int, max_calls: int = 100000) -> list[dict]:
        """Feed `x` like the scheduler would: general_work with up to `chunk` available items from the read pointer, advance
        by what the block consumed, until the stream is exhausted or the block has gone inactive.  Returns the status after every call
        in which something changed state / events."""
        pos, log = 0, []
        for _ in range(max_calls):
            avail = min(chunk, len(x) - pos)
            if avail <= 0:
                break
            _, consumed = self.work(x[pos:pos + avail])
            pos += consumed
            st = self.status()
            st["pos"] = pos
            log.append(st)
            if not st["active"]:
                break
        return log
