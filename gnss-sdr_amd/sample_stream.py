"""Python face of gsh_stream_* (include/gnss_sdr_hip.h): the device-resident IF sample ring, for tests and bench."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import GSH_ITEM_BYTE, GSH_ITEM_GR_COMPLEX, GSH_ITEM_SHORT, check, fptr

ITEM_TYPES = {"gr_complex": GSH_ITEM_GR_COMPLEX, "ishort": GSH_ITEM_SHORT, "cshort": GSH_ITEM_SHORT, "ibyte": GSH_ITEM_BYTE, "cbyte": GSH_ITEM_BYTE}
_NP = {GSH_ITEM_GR_COMPLEX: np.complex64, GSH_ITEM_SHORT: np.int16, GSH_ITEM_BYTE: np.int8}


class SampleStream:
    def __init__(self, capacity_samples: int, max_window_samples: int, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.device = device
        check(self._lib.gsh_stream_create(device, capacity_samples, max_window_samples, C.byref(self._h)))

    def close(self):
        if self._h and getattr(self, "_owned", True):
            self._lib.gsh_stream_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, items: np.ndarray, item_type: str = "gr_complex", inverted_spectrum: bool = False) -> int:
        """Append samples held in host memory (complex64 [n], or int16 / int8 [n, 2] interleaved I,Q).  Returns the absolute
        index of the first one."""
        t = ITEM_TYPES[item_type]
        items = np.asarray(items)
        if t == GSH_ITEM_GR_COMPLEX and not np.iscomplexobj(items):
            items = np.ascontiguousarray(items, np.float32).reshape(-1).view(np.complex64)   # interleaved float32 I,Q as read from a file
        a = np.ascontiguousarray(items, _NP[t])
        n = a.size if t == GSH_ITEM_GR_COMPLEX else a.size // 2
        first = C.c_uint64(0)
        check(self._lib.gsh_stream_push(self._h, C.c_void_p(a.ctypes.data), n, t, int(inverted_spectrum), C.byref(first)))
        return int(first.value)

    def push_device(self, device_ptr: int, n: int, item_type: str = "gr_complex", inverted_spectrum: bool = False, hip_stream: int = 0) -> int:
        first = C.c_uint64(0)
        check(self._lib.gsh_stream_push_device(self._h, C.c_void_p(device_ptr), n, ITEM_TYPES[item_type], int(inverted_spectrum),
                                               C.c_void_p(hip_stream) if hip_stream else None, C.byref(first)))
        return int(first.value)

    def push_async(self, items: np.ndarray, item_type: str = "gr_complex", inverted_spectrum: bool = False) -> int:
        """gsh_stream_push_async: copy + conversion queued on the ring's stream, returns at once.  `items` (numpy array, or a host pointer
        given as (ptr, n)) must stay untouched until wait() or two further pushes."""
        first = C.c_uint64(0)
        if isinstance(items, tuple):
            ptr, n = items
        else:
            ptr, n = items.ctypes.data, (items.size // 2 if item_type != "gr_complex" or items.dtype != np.complex64 else items.size)
            self._keep_async = getattr(self, "_keep_async", [])[-2:] + [items]
        check(self._lib.gsh_stream_push_async(self._h, C.c_void_p(ptr), n, ITEM_TYPES[item_type], int(inverted_spectrum), C.byref(first)))
        return int(first.value)

    def push_pinned(self, items: np.ndarray, item_type: str = "gr_complex", inverted_spectrum: bool = False) -> int:
        """gsh_stream_push_pinned: `items` is page-locked for the call (gsh_host_register on its pages) and handed to the DMA engine as it lies --
        gr_complex items straight into their ring positions; returns when the array may be reused."""
        t = ITEM_TYPES[item_type]
        a = np.ascontiguousarray(items, _NP[t])
        n = a.size if t == GSH_ITEM_GR_COMPLEX else a.size // 2
        first = C.c_uint64(0)
        if n == 0:
            check(self._lib.gsh_stream_push_pinned(self._h, None, 0, t, int(inverted_spectrum), C.byref(first)))
            return int(first.value)
        lo = a.ctypes.data & ~4095
        hi = (a.ctypes.data + a.nbytes + 4095) & ~4095
        check(self._lib.gsh_host_register(self.device, C.c_void_p(lo), hi - lo))
        try:
            check(self._lib.gsh_stream_push_pinned(self._h, C.c_void_p(a.ctypes.data), n, t, int(inverted_spectrum), C.byref(first)))
        finally:
            check(self._lib.gsh_host_unregister(C.c_void_p(lo)))
        return int(first.value)

    def wait(self) -> None:
        check(self._lib.gsh_stream_wait(self._h))

    def seek(self, next_index: int) -> None:
        check(self._lib.gsh_stream_seek(self._h, int(next_index)))

    def range(self):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        check(self._lib.gsh_stream_range(self._h, C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def read(self, index: int, n: int) -> np.ndarray:
        out = np.empty(n, np.complex64)
        check(self._lib.gsh_stream_read(self._h, index, n, fptr(out)))
        return out


def convert_samples_device(device: int, src_ptr: int, item_type: str, dst_ptr: int, n: int, inverted_spectrum: bool = False, hip_stream: int = 0) -> None:
    check(_lib.load().gsh_convert_samples_device(device, C.c_void_p(src_ptr), ITEM_TYPES[item_type], int(inverted_spectrum), C.c_void_p(dst_ptr), n,
                                                 C.c_void_p(hip_stream) if hip_stream else None))


def direct_resample_device(device: int, src_ptr: int, in0: int, n_in: int, fs_in: float, fs_out: float, out0: int, dst_ptr: int, max_out: int,
                           hip_stream: int = 0):
    """direct_resampler_conditioner_cc on the device (gsh_direct_resample_device).  -> (n_out, n_in_consumed)"""
    n_out, n_cons = C.c_uint64(0), C.c_uint64(0)
    check(_lib.load().gsh_direct_resample_device(device, C.c_void_p(src_ptr), in0, n_in, fs_in, fs_out, out0, C.c_void_p(dst_ptr), max_out,
                                                 C.byref(n_out), C.byref(n_cons), C.c_void_p(hip_stream) if hip_stream else None))
    return int(n_out.value), int(n_cons.value)


class FirFilter:
    """gsh_fir_*: frequency-translating decimating FIR filter with stream history (freq_xlating_fir_filter_ccf / fir_filter_ccf on the device)."""
    KINDS = {"gr_complex": 0, "float": 1, "short": 2, "byte": 3}

    def __init__(self, taps, decimation: int = 1, center_freq_hz: float = 0.0, sampling_freq_hz: float = 1.0, input_kind: str = "gr_complex", device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        t = np.ascontiguousarray(taps, np.float32)
        self.decimation = decimation
        check(self._lib.gsh_fir_create(device, fptr(t), len(t), decimation, center_freq_hz, sampling_freq_hz, self.KINDS[input_kind], C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gsh_fir_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_device(self, in_ptr: int, n_in: int, out_ptr: int, max_out: int, hip_stream: int = 0) -> int:
        n_out = C.c_uint64(0)
        check(self._lib.gsh_fir_process_device(self._h, C.c_void_p(in_ptr), n_in, C.c_void_p(out_ptr), max_out, C.byref(n_out),
                                               C.c_void_p(hip_stream) if hip_stream else None))
        return int(n_out.value)


def firdes_low_pass(gain: float, sampling_freq: float, cutoff_freq: float, transition_width: float) -> np.ndarray:
    """gr::filter::firdes::low_pass with the default Hamming window, as restated in host/hip_acq_resampler.h (GNU Radio is not
    vendored: published algorithm, parity unpinned)."""
    ntaps = int(53.0 * sampling_freq / (22.0 * transition_width))
    if ntaps % 2 == 0:
        ntaps += 1
    n = np.arange(ntaps)
    w = (0.54 - 0.46 * np.cos(2.0 * np.pi * n / (ntaps - 1))).astype(np.float32)
    m = (ntaps - 1) // 2
    k = n - m
    fw = 2.0 * np.pi * cutoff_freq / sampling_freq
    with np.errstate(invalid="ignore", divide="ignore"):
        taps = np.where(k == 0, fw / np.pi, np.sin(k * fw) / (k * np.pi)) * w
    taps = taps.astype(np.float32)
    fmax = float(taps[m]) + 2.0 * float(np.sum(taps[m + 1:].astype(np.float64)))
    return (taps.astype(np.float64) * (gain / fmax)).astype(np.float32)


def acquisition_resampler_design(fs: int, acq_fs: float):
    """(decimation, decimated rate, taps, latency) as GNSSFlowgraph sets the acquisition resampler up (gnss_flowgraph.cc:1165-1211)."""
    if not acq_fs < fs:
        return 1, float(fs), np.zeros(0, np.float32), 0
    decimation = int(np.floor(fs / acq_fs))
    while fs % decimation > 0:
        decimation -= 1
    if decimation <= 1:
        return 1, float(fs), np.zeros(0, np.float32), 0
    dec_fs = fs / decimation
    taps = firdes_low_pass(1.0, fs, dec_fs / 2.1, dec_fs / 2)
    return decimation, dec_fs, taps, (len(taps) - 1) // 2


class PulseBlanking:
    """gsh_pb_*: pulse_blanking_cc on the device (pulse_blanking_cc.cc:33-106)."""

    def __init__(self, pfa: float = 0.04, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_pb_create(device, pfa, length, n_segments_est, n_segments_reset, C.byref(self._h)))
        self.length = length

    def close(self):
        if self._h:
            self._lib.gsh_pb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def threshold(self) -> float:
        return float(self._lib.gsh_pb_threshold(self._h))

    def process_device(self, d_in: int, n_items: int, d_out: int) -> int:
        """One general_work call over n_items resident samples; returns how many were consumed (= produced)."""
        done = C.c_uint64(0)
        check(self._lib.gsh_pb_process_device(self._h, C.c_void_p(d_in), n_items, C.c_void_p(d_out), C.byref(done)))
        return int(done.value)

    def state(self):
        noise, n, last = C.c_float(0.0), C.c_int32(0), C.c_int32(0)
        check(self._lib.gsh_pb_get_state(self._h, C.byref(noise), C.byref(n), C.byref(last)))
        return float(noise.value), int(n.value), bool(last.value)


class NotchFilter:
    """gsh_notch_*: Notch (notch_cc.cc:33-140; n_segments_coeff = 0) or NotchLite (notch_lite_cc.cc:30-150; n_segments_coeff >= 1) on the device."""

    def __init__(self, pfa: float = 0.001, p_c_factor: float = 0.9, length: int = 32, n_segments_est: int = 12500, n_segments_reset: int = 5000000,
                 n_segments_coeff: int = 0, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.gsh_notch_create(device, pfa, p_c_factor, length, n_segments_est, n_segments_reset, n_segments_coeff, C.byref(self._h)))
        self.length = length

    def close(self):
        if self._h:
            self._lib.gsh_notch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def threshold(self) -> float:
        return float(self._lib.gsh_notch_threshold(self._h))

    def process_device(self, d_in: int, n_items: int, d_out: int) -> int:
        """One general_work call over n_items resident items (item 0: the sample in front of the first one processed); returns how many were
        consumed (= produced; out[k] is the filtered in[k + 1])."""
        done = C.c_uint64(0)
        check(self._lib.gsh_notch_process_device(self._h, C.c_void_p(d_in), n_items, C.c_void_p(d_out), C.byref(done)))
        return int(done.value)

    def state(self) -> dict:
        noise, n, fs, nc = C.c_float(0.0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        lo, z0 = (C.c_float * 2)(), (C.c_float * 2)()
        check(self._lib.gsh_notch_get_state(self._h, C.byref(noise), C.byref(n), C.byref(fs), lo, C.byref(nc), z0))
        return dict(noise_pow_est=float(noise.value), n_segments=int(n.value), filter_state=bool(fs.value), last_out=complex(lo[0], lo[1]),
                    n_segments_coeff=int(nc.value), z0=complex(z0[0], z0[1]))


class StreamGroup:
    """gsh_stream_group_*: one block replicated into the sample rings of several GPUs over RCCL (one process per GPU: from_rank; one process
    driving several GPUs: local)."""
    MODES = {"broadcast": 0, "scatter_allgather": 1}
    FORCE_RCCL = 0x100   # GSH_GROUP_FORCE_RCCL: a group of one builds its communicator and runs the mode's collectives all the same

    def __init__(self, handle, lib):
        self._h, self._lib = handle, lib

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().gsh_comm_unique_id(buf))
        return buf.raw

    class _Op(C.Structure):
        _fields_ = [("op", C.c_int32), ("peer", C.c_int32), ("phase", C.c_int32), ("src_buf", C.c_int32), ("dst_buf", C.c_int32),
                    ("src_offset", C.c_uint64), ("dst_offset", C.c_uint64), ("bytes", C.c_uint64)]

    OPS = {0: "broadcast", 1: "send", 2: "recv", 3: "allgather"}
    BUFS = {0: None, 1: "stage", 2: "piece"}

    @classmethod
    def plan(cls, nbytes: int, world: int, rank: int, mode: str = "broadcast"):
        """gsh_stream_group_plan: (padded_bytes, [dict(op, peer, phase, src_buf, dst_buf, src_offset, dst_offset, bytes)]) -- the operations a push of
        `nbytes` raw bytes issues on `rank`, the list the engine itself walks.  Needs no GPU."""
        L = _lib.load()
        n, padded = C.c_int(0), C.c_uint64(0)
        check(L.gsh_stream_group_plan(nbytes, world, rank, cls.MODES[mode], None, 0, C.byref(n), C.byref(padded)))
        ops = (cls._Op * max(n.value, 1))()
        check(L.gsh_stream_group_plan(nbytes, world, rank, cls.MODES[mode], ops, n.value, C.byref(n), C.byref(padded)))
        return int(padded.value), [dict(op=cls.OPS[o.op], peer=int(o.peer), phase=int(o.phase), src_buf=cls.BUFS[o.src_buf], dst_buf=cls.BUFS[o.dst_buf],
                                        src_offset=int(o.src_offset), dst_offset=int(o.dst_offset), bytes=int(o.bytes)) for o in ops[:n.value]]

    @staticmethod
    def library() -> str:
        """File name of the collective library the engine has loaded (gsh_comm_library): the system's librccl, or what GSH_RCCL_LIBRARY names."""
        buf = C.create_string_buffer(1024)
        check(_lib.load().gsh_comm_library(buf, 1024))
        return buf.value.decode()

    @classmethod
    def from_rank(cls, device: int, rank: int, world: int, unique_id: bytes | None, capacity_samples: int, max_window_samples: int, mode: str = "broadcast",
                  force_rccl: bool = False):
        L = _lib.load()
        h = C.c_void_p()
        idb = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        check(L.gsh_stream_group_create_rank(device, rank, world, idb, capacity_samples, max_window_samples,
                                             cls.MODES[mode] | (cls.FORCE_RCCL if force_rccl else 0), C.byref(h)))
        return cls(h, L)

    @classmethod
    def local(cls, devices, capacity_samples: int, max_window_samples: int, mode: str = "broadcast", force_rccl: bool = False):
        L = _lib.load()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        check(L.gsh_stream_group_create(arr, len(devices), capacity_samples, max_window_samples,
                                        cls.MODES[mode] | (cls.FORCE_RCCL if force_rccl else 0), C.byref(h)))
        return cls(h, L)

    def rccl_info(self) -> dict:
        """{'ranks': ranks of the communicator the group built (0: RCCL untouched), 'version': ncclGetVersion code, 'collectives': RCCL calls issued}"""
        ranks, ver, calls = C.c_int32(0), C.c_int32(0), C.c_uint64(0)
        check(self._lib.gsh_stream_group_rccl_info(self._h, C.byref(ranks), C.byref(ver), C.byref(calls)))
        return dict(ranks=int(ranks.value), version=int(ver.value), collectives=int(calls.value))

    def close(self):
        if self._h:
            self._lib.gsh_stream_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        return self._lib.gsh_stream_group_size(self._h)

    def ring(self, local_index: int = 0) -> "SampleStream":
        """The group's ring as a SampleStream view (owned by the group: closing the view does nothing)."""
        s = SampleStream.__new__(SampleStream)
        s._lib = self._lib
        s._h = C.c_void_p(self._lib.gsh_stream_group_ring(self._h, local_index))
        s._owned = False
        s._group = self
        return s

    def push_device(self, device_ptr: int | None, n: int, item_type: str = "ibyte", inverted_spectrum: bool = False) -> int:
        first = C.c_uint64(0)
        check(self._lib.gsh_stream_group_push_device(self._h, C.c_void_p(device_ptr) if device_ptr else None, n, ITEM_TYPES[item_type],
                                                     int(inverted_spectrum), C.byref(first)))
        return int(first.value)

    def push(self, items: np.ndarray | None, n: int, item_type: str = "ibyte", inverted_spectrum: bool = False) -> int:
        first = C.c_uint64(0)
        ptr = C.c_void_p(items.ctypes.data) if items is not None else None
        check(self._lib.gsh_stream_group_push(self._h, ptr, n, ITEM_TYPES[item_type], int(inverted_spectrum), C.byref(first)))
        return int(first.value)

    def wait(self) -> None:
        check(self._lib.gsh_stream_group_wait(self._h))
