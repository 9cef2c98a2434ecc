"""Python face of the device-resident DLL/PLL loop (gsh_trk_*), for tests and bench.

``TrackingLoop`` owns one handle: n_channels channels of one signal type sharing an IF stream.  ``start`` is the
reference's ``start_tracking`` + pull-in hand-over (dll_pll_veml_tracking.cc:796-866, 1949-1973), ``run`` executes
n code periods of every started channel in one launch and returns the per-period records (what ``log_data`` dumps).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TrkConf, TrkEpoch, check, fptr


def trk_conf(**kw) -> TrkConf:
    """gsh_trk_conf with Dll_Pll_Conf's defaults (src/algorithms/tracking/libs/dll_pll_conf.h:33-90)."""
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
             high_dyn=0, smoother_length=10,
             # the state-2 fail-safe (trk.cc:2000-2007; off unless asked for: the reference has no switch, its adapters get it switched on) and the
             # experimental Doppler correction (trk.cc:1326-1346, Dll_Pll_Conf default false)
             enable_bit_sync_time_limit=0, bit_synchronization_time_limit_s=20, enable_doppler_correction=0)
    d.update(kw)
    for k, v in d.items():
        setattr(c, k, v)
    return c


class TrackingLoop:
    def __init__(self, conf: TrkConf, n_channels: int, max_code_length: int, device: int = 0):
        self._lib = _lib.load()
        self.conf = conf
        self.n_channels = n_channels
        self._h = C.c_void_p()
        self._keep = {}
        check(self._lib.gsh_trk_create(device, C.byref(conf), n_channels, max_code_length, C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gsh_trk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream_host(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, np.complex64)
        check(self._lib.gsh_trk_set_stream_host(self._h, fptr(x), len(x)))

    def set_stream_device(self, device_ptr: int, n_samples: int, keepalive=None) -> None:
        self._keep["stream"] = keepalive
        check(self._lib.gsh_trk_set_stream_device(self._h, C.c_void_p(device_ptr), n_samples))

    def set_stream_ring(self, ring) -> None:
        """Follow a live SampleStream ring: positions become absolute sample indices; run() processes what is resident."""
        self._keep["stream"] = ring
        check(self._lib.gsh_trk_set_stream_ring(self._h, ring._h if ring is not None else None))

    def start(self, channel: int, code: np.ndarray, start_sample: int, acq_sample_stamp: int, acq_carrier_doppler_hz: float,
              data_code: np.ndarray | None = None, pull_in_over: bool = False) -> None:
        """pull_in_over: the pull-in transitory was over at the hand-over call already (GSH_TRK_START_PULL_IN_OVER, see gsh_trk_pull_in_over)."""
        code = np.ascontiguousarray(code, np.float32)
        dc = None
        if data_code is not None:
            data_code = np.ascontiguousarray(data_code, np.float32)
            dc = fptr(data_code)
        check(self._lib.gsh_trk_start_flags(self._h, channel, fptr(code), dc, len(code), int(start_sample), int(acq_sample_stamp),
                                            float(acq_carrier_doppler_hz), 0.0, 1 if pull_in_over else 0))

    def run(self, n_epochs: int, want_records: bool = True):
        """-> (records[channel][epoch] as lists of TrkEpoch, epochs_done per channel)"""
        done = (C.c_int32 * self.n_channels)()
        if want_records:
            rec = (TrkEpoch * (self.n_channels * n_epochs))()
            check(self._lib.gsh_trk_run(self._h, n_epochs, rec, done))
            out = [[rec[ch * n_epochs + e] for e in range(done[ch])] for ch in range(self.n_channels)]
        else:
            check(self._lib.gsh_trk_run(self._h, n_epochs, None, done))
            out = None
        return out, list(done)

    # ---- live mode (gsh_trk_live_*): the loop stays resident and follows the ring; records are read from host memory without a device call
    def live_configure(self, idle_timeout_us: int = 200, residency_us: int = 5000) -> None:
        check(self._lib.gsh_trk_live_configure(self._h, idle_timeout_us, residency_us))

    def live_begin(self) -> None:
        check(self._lib.gsh_trk_live_begin(self._h))

    def live_in_flight(self) -> int:
        n = C.c_int32(0)
        check(self._lib.gsh_trk_live_in_flight(self._h, C.byref(n)))
        return n.value

    def live_take(self, channel: int, max_records: int = 64, limit_end: int = 2**64 - 1):
        """-> (records, pending, next_window, active)"""
        rec = (TrkEpoch * max_records)()
        n, pending, active = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        nw = C.c_uint64(0)
        check(self._lib.gsh_trk_live_take(self._h, channel, C.c_uint64(limit_end), max_records, rec, C.byref(n), C.byref(pending), C.byref(nw), C.byref(active), None))
        out = []
        for i in range(n.value):  # copies: the ctypes array is reused by nobody, but records outlive it this way
            r = TrkEpoch()
            C.memmove(C.byref(r), C.byref(rec[i]), C.sizeof(TrkEpoch))
            out.append(r)
        return out, pending.value, nw.value, active.value

    def live_quiesce(self) -> None:
        check(self._lib.gsh_trk_live_quiesce(self._h))

    def set_split(self, work_groups_per_channel: int) -> None:
        """gsh_trk_set_split: work-groups that share every window of a channel in launched runs (1 = off)."""
        check(self._lib.gsh_trk_set_split(self._h, work_groups_per_channel))

    def time_run(self, n_epochs: int, reps: int = 5) -> float:
        ms = C.c_float(0.0)
        check(self._lib.gsh_trk_time_run(self._h, n_epochs, reps, C.byref(ms)))
        return ms.value


def write_dump(path: str, conf: TrkConf, prn: int, records, append: bool = False, tow_ms=None, wn=None) -> None:
    """Tracking dump file in the block's binary layout (log_data, trk.cc:1599-1702: 108 bytes per logged period) from a list of TrkEpoch
    records; tow_ms / wn: one value per record (the TOW hand-back of trk.cc:1921-1935) or None for zeros."""
    arr = (TrkEpoch * len(records))(*records)
    tow = (C.c_uint64 * len(records))(*[int(v) for v in tow_ms]) if tow_ms is not None else None
    week = (C.c_uint32 * len(records))(*[int(v) for v in wn]) if wn is not None else None
    check(_lib.load().gsh_trk_write_dump(str(path).encode(), int(append), C.byref(conf), prn, arr, len(records), tow, week))


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
