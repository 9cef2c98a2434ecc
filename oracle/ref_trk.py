"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/_ref/libgnsssdr_ref_trk.so: the REFERENCE's own tracking chain (TrackingInterface
adapters -> Dll_Pll_Conf -> dll_pll_veml_tracking block -> Cpu_Multicorrelator_Real_Codes + discriminators + loop filters + lock detectors),
compiled from /root/reference by oracle/Makefile (see oracle/ref_trk_api.cc) and driven through general_work like the GNU Radio scheduler.

Used to PIN oracle/gnss_oracle_loop.c at block level (tests/test_oracle_loop_pinned.py) and as the checker of the HIP tracking block.
The product never imports this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libgnsssdr_ref_trk.so")


class Output(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("fs", "prompt_i", "prompt_q", "cn0_db_hz", "carrier_doppler_hz", "carrier_phase_rads", "code_phase_samples")] + [
        ("tracking_sample_counter", C.c_uint64),
        ("flag_valid_symbol_output", C.c_int32), ("correlation_length_ms", C.c_int32), ("flag_pll_180_deg_phase_locked", C.c_int32), ("prn", C.c_int32),
        ("state", C.c_int32), ("current_prn_length_samples", C.c_int32), ("n_correlator_taps", C.c_int32), ("cn0_estimation_counter", C.c_int32),
        ("carrier_lock_fail_counter", C.c_int32), ("code_lock_fail_counter", C.c_int32)] + [
        (k, C.c_double) for k in ("code_freq_chips", "rem_code_phase_samples", "rem_code_phase_chips", "acc_carrier_phase_rad", "carrier_lock_test",
                                  "carr_phase_error_hz", "carr_freq_error_hz", "carr_error_filt_hz", "code_error_chips", "code_error_filt_chips",
                                  "carrier_phase_step_rad", "code_phase_step_chips", "carrier_phase_rate_step_rad", "code_phase_rate_step_chips",
                                  "current_correlation_time_s")] + [
        ("rem_carr_phase_rad", C.c_float), ("corr", C.c_float * 10), ("prompt_data", C.c_float * 2), ("accu", C.c_float * 10),
        ("p_data_accu", C.c_float * 2), ("n_events", C.c_int32), ("events", C.c_int32 * 16),
        ("tow_at_current_symbol_ms", C.c_uint64)]

    def as_dict(self) -> dict:
        d = {}
        for k, _ in self._fields_:
            v = getattr(self, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        d["events"] = d["events"][:self.n_events]
        return d


class ConfOut(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("fs_in", "carrier_lock_th", "signal_carrier_freq", "code_period", "code_chip_rate", "bs_dominance_ratio")] + [
        (k, C.c_float) for k in ("pll_bw_hz", "dll_bw_hz", "fll_bw_hz", "pll_bw_narrow_hz", "dll_bw_narrow_hz", "early_late_space_chips",
                                 "very_early_late_space_chips", "early_late_space_narrow_chips", "very_early_late_space_narrow_chips", "slope", "spc",
                                 "y_intercept", "cn0_smoother_alpha", "carrier_lock_test_smoother_alpha", "bs_min_prompt_mag")] + [
        (k, C.c_uint32) for k in ("pull_in_time_s", "bit_synchronization_time_limit_s", "vector_length", "smoother_length")] + [
        (k, C.c_int32) for k in ("pll_filter_order", "dll_filter_order", "fll_filter_order", "extend_correlation_symbols", "cn0_samples",
                                 "cn0_smoother_samples", "carrier_lock_test_smoother_samples", "cn0_min", "max_code_lock_fail", "max_carrier_lock_fail",
                                 "bs_stable_best_required", "bs_min_events_for_lock",
                                 "enable_fll_pull_in", "enable_fll_steady_state", "track_pilot", "carrier_aiding", "high_dyn", "bs_use_phase_dot_detector",
                                 "code_length_chips", "code_samples_per_chip", "symbols_per_bit", "secondary", "veml", "cloop", "use_histogram_bit_sync",
                                 "interchange_iq", "secondary_code_length", "data_secondary_code_length", "correlation_length_ms", "n_correlator_taps", "enable_doppler_correction")] + [
        ("secondary_code", C.c_char * 256), ("data_secondary_code", C.c_char * 256), ("system", C.c_char), ("signal", C.c_char * 3)]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        for k in ("secondary_code", "data_secondary_code", "signal", "system"):
            d[k] = d[k].decode() if isinstance(d[k], bytes) else d[k]
        return d


_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.reftrk_create.restype = C.c_void_p
        L.reftrk_create.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int]
        L.reftrk_destroy.argtypes = [C.c_void_p]
        L.reftrk_set_acquisition.argtypes = [C.c_void_p, C.c_char, C.c_char_p, C.c_uint32, C.c_double, C.c_double, C.c_uint64]
        L.reftrk_start_tracking.argtypes = [C.c_void_p]
        L.reftrk_stop_tracking.argtypes = [C.c_void_p]
        L.reftrk_set_doppler_correction.argtypes = [C.c_void_p, C.c_int]
        L.reftrk_forecast.argtypes = [C.c_void_p, C.c_int]
        L.reftrk_nitems_read.restype = C.c_uint64
        L.reftrk_nitems_read.argtypes = [C.c_void_p]
        L.reftrk_general_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(Output)]
        L.reftrk_clear_events.argtypes = [C.c_void_p]
        L.reftrk_get_conf.argtypes = [C.c_void_p, C.POINTER(ConfOut)]
        L.reftrk_get_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


class RefTrackingChannel:
    """One reference tracking adapter + block + Gnss_Synchro, built from configuration properties the way GNSSBlockFactory::GetTrkBlock does
    (gnss_block_factory.cc:582-700: constructor(configuration, role, in_streams, out_streams))."""

    def __init__(self, implementation: str, props: dict, role: str = "Tracking"):
        L = lib()
        keys = (C.c_char_p * len(props))(*[k.encode() for k in props])
        vals = (C.c_char_p * len(props))(*[str(v).encode() for v in props.values()])
        self.h = L.reftrk_create(implementation.encode(), role.encode(), keys, vals, len(props))
        if not self.h:
            raise RuntimeError(f"reference tracking adapter {implementation} could not be built")

    def close(self):
        if self.h:
            lib().reftrk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def conf(self) -> dict:
        c = ConfOut()
        lib().reftrk_get_conf(self.h, C.byref(c))
        return c.as_dict()

    def set_acquisition(self, system: str, signal: str, prn: int, acq_delay_samples: float, acq_doppler_hz: float, acq_samplestamp: int):
        lib().reftrk_set_acquisition(self.h, system.encode(), signal.encode(), prn, acq_delay_samples, acq_doppler_hz, acq_samplestamp)

    def start_tracking(self):
        lib().reftrk_start_tracking(self.h)

    def stop_tracking(self):
        lib().reftrk_stop_tracking(self.h)

    def set_doppler_correction(self, on: bool):
        """the block's d_trk_parameters.enable_doppler_correction (no configuration key exists for it in the reference)"""
        lib().reftrk_set_doppler_correction(self.h, 1 if on else 0)

    def forecast(self, noutput: int = 1) -> int:
        return lib().reftrk_forecast(self.h, noutput)

    def nitems_read(self) -> int:
        return lib().reftrk_nitems_read(self.h)

    def codes(self):
        n = self.conf()["code_length_chips"] * self.conf()["code_samples_per_chip"]
        a = np.zeros(n, np.float32)
        b = np.zeros(n, np.float32)
        r = lib().reftrk_get_codes(self.h, a.ctypes.data, b.ctypes.data, n)
        assert r == n, r
        return a, b

    def work(self, x: np.ndarray):
        """one general_work call; x: the samples available from the read pointer.  -> (produced, consumed, Output dict)"""
        x = np.ascontiguousarray(x, np.complex64)
        consumed = C.c_int(0)
        out = Output()
        r = lib().reftrk_general_work(self.h, x.ctypes.data, len(x), C.byref(consumed), C.byref(out))
        return r, consumed.value, out.as_dict()

    def run(self, x: np.ndarray, n_periods: int, available: int | None = None):
        """feed the stream from the block's current read position for n_periods code periods (general_work calls in state >= 2; the
        pull-in call that only aligns the stream is not counted); every call offers `available` samples (default: forecast(1)).
        -> list of Output dicts, one per period (item fields are zero when the block produced no Gnss_Synchro in that call: it only
        does once telemetry symbols come out, trk.cc:2285-2327)"""
        outs = []
        pos = self.nitems_read()
        need = available or self.forecast(1)
        while len(outs) < n_periods and pos + need <= len(x):
            r, c, o = self.work(x[pos:pos + need])
            o["produced"], o["consumed"], o["read_pos"] = r, c, pos
            pos += c
            if o["state"] == 0:      # loss of lock (or never started)
                o["lost"] = True
                outs.append(o)
                break
            if o["corr"][2] != 0.0 or o["corr"][3] != 0.0 or r > 0 or outs:
                outs.append(o)
        return outs
