"""ctypes binding of the C ABI in include/gnss_sdr_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
fallback of any kind: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSH_LIB_PATH") or os.path.join(_HERE, "libgnss_sdr_hip.so")  # override: kernel-tuning experiments only

GSH_MAX_TAPS = 8
GSH_OK = 0
GSH_ITEM_GR_COMPLEX, GSH_ITEM_SHORT, GSH_ITEM_BYTE = 0, 1, 2
ERR_NAMES = {1: "GSH_ERR_INVALID", 2: "GSH_ERR_NO_DEVICE", 3: "GSH_ERR_HIP", 4: "GSH_ERR_STATE", 5: "GSH_ERR_UNSUPPORTED"}


class GshError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {text}")
        self.code = code


class CorrJob(C.Structure):
    """gsh_corr_job (80 bytes)."""
    _fields_ = [
        ("sample_offset", C.c_uint64),
        ("n_samples", C.c_int32),
        ("code_slot", C.c_int32),
        ("rem_carr_phase_rad", C.c_float),
        ("phase_step_rad", C.c_float),
        ("phase_rate_step_rad", C.c_float),
        ("rem_code_phase_chips", C.c_float),
        ("code_phase_step_chips", C.c_float),
        ("code_phase_rate_step_chips", C.c_float),
        ("n_taps", C.c_int32),
        ("high_dyn", C.c_int32),
        ("shifts_chips", C.c_float * GSH_MAX_TAPS),
    ]


class Corr16Job(C.Structure):
    """gsh_corr16_job (72 bytes)."""
    _fields_ = [
        ("sample_offset", C.c_uint64),
        ("n_samples", C.c_int32),
        ("code_slot", C.c_int32),
        ("rem_carr_phase_rad", C.c_float),
        ("phase_step_rad", C.c_float),
        ("rem_code_phase_chips", C.c_float),
        ("code_phase_step_chips", C.c_float),
        ("n_taps", C.c_int32),
        ("reserved", C.c_int32),
        ("shifts_chips", C.c_float * GSH_MAX_TAPS),
    ]


class AcqConf(C.Structure):
    """gsh_acq_conf."""
    _fields_ = [
        ("fs_in", C.c_int64),
        ("fft_size", C.c_uint32),
        ("effective_fft_size", C.c_uint32),
        ("consumed_samples", C.c_uint32),
        ("num_doppler_bins", C.c_uint32),
        ("doppler_max", C.c_int32),
        ("doppler_step", C.c_int32),
        ("doppler_center", C.c_int32),
        ("doppler_bias", C.c_int32),
        ("samples_per_chip", C.c_uint32),
        ("samples_per_code", C.c_float),
        ("bit_transition_flag", C.c_int32),
        ("use_cfar", C.c_int32),
        ("max_prn", C.c_uint32),
        ("no_grid", C.c_int32),
        ("transform_path", C.c_int32),
        ("num_doppler_bins_step2", C.c_uint32),
        ("doppler_step2", C.c_float),
        ("fold", C.c_uint32),
    ]


class TrkConf(C.Structure):
    """gsh_trk_conf."""
    _fields_ = [
        ("fs_in", C.c_double), ("code_chip_rate", C.c_double), ("signal_carrier_freq", C.c_double), ("cfo_frequency_hz", C.c_double),
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
                ("high_dyn", C.c_int32), ("smoother_length", C.c_uint32), ("bit_synchronization_time_limit_s", C.c_uint32), ("enable_doppler_correction", C.c_int32),
    ]


class TrkEpoch(C.Structure):
    """gsh_trk_epoch."""
    _fields_ = [
        ("sample_counter", C.c_uint64), ("prn_length_samples", C.c_int32), ("flags", C.c_int32),
        ("corr", C.c_float * 10), ("prompt_data", C.c_float * 2), ("rem_carr_phase_rad", C.c_float), ("cn0_db_hz", C.c_float),
        ("carrier_doppler_hz", C.c_double), ("code_freq_chips", C.c_double), ("carr_phase_error_hz", C.c_double),
        ("carr_freq_error_hz", C.c_double), ("carr_error_filt_hz", C.c_double), ("code_error_chips", C.c_double),
        ("code_error_filt_chips", C.c_double), ("rem_code_phase_samples", C.c_double), ("acc_carrier_phase_rad", C.c_double), ("carrier_lock_test", C.c_double),
                ("state", C.c_int32), ("symbol_flags", C.c_int32), ("p_data_accu", C.c_float * 2),
                ("carrier_phase_rate_step_rad", C.c_double), ("code_phase_rate_step_chips", C.c_double),
                ("accu", C.c_float * 10),
    ]


class AcqResult(C.Structure):
    """gsh_acq_result."""
    _fields_ = [
        ("index_time", C.c_uint32),
        ("index_doppler", C.c_uint32),
        ("doppler_hz", C.c_int32),
        ("acq_delay_samples", C.c_float),
        ("peak", C.c_float),
        ("input_power", C.c_float),
        ("second_peak", C.c_float),
        ("test_statistics", C.c_float),
    ]


# every symbol include/gnss_sdr_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I16 = C.POINTER(C.c_int16)
SYMBOLS = {
    "gsh_abi_version": (C.c_int, []),
    "gsh_device_count": (C.c_int, []),
    "gsh_probe_read_bandwidth": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "gsh_last_error": (C.c_char_p, []),
    "gsh_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "gsh_mcorr_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gsh_mcorr_destroy": (None, [_P]),
    "gsh_mcorr_init": (C.c_int, [_P, C.c_int, C.c_int]),
    "gsh_mcorr_set_local_code_and_taps": (C.c_int, [_P, C.c_int, _F, _F]),
    "gsh_mcorr_set_input_output_vectors": (C.c_int, [_P, _F, _F]),
    "gsh_mcorr_set_high_dynamics_resampler": (C.c_int, [_P, C.c_int]),
    "gsh_mcorr_carrier_wipeoff_multicorrelator_resampler": (C.c_int, [_P] + [C.c_float] * 6 + [C.c_int]),
    "gsh_mcorr_carrier_wipeoff_multicorrelator_resampler6": (C.c_int, [_P] + [C.c_float] * 5 + [C.c_int]),
    "gsh_mcorr_free": (C.c_int, [_P]),
    "gsh_mcorr16_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gsh_mcorr16_destroy": (None, [_P]),
    "gsh_mcorr16_init": (C.c_int, [_P, C.c_int, C.c_int]),
    "gsh_mcorr16_set_local_code_and_taps": (C.c_int, [_P, C.c_int, _I16, _F]),
    "gsh_mcorr16_set_input_output_vectors": (C.c_int, [_P, _I16, _I16]),
    "gsh_mcorr16_carrier_wipeoff_multicorrelator_resampler": (C.c_int, [_P] + [C.c_float] * 4 + [C.c_int]),
    "gsh_mcorr16_free": (C.c_int, [_P]),
    "gsh_bank16_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gsh_bank16_destroy": (None, [_P]),
    "gsh_bank16_set_code": (C.c_int, [_P, C.c_int, _I16, C.c_int]),
    "gsh_bank16_set_stream_host": (C.c_int, [_P, _I16, C.c_uint64]),
    "gsh_bank16_set_stream_device": (C.c_int, [_P, _P, C.c_uint64]),
    "gsh_bank16_correlate": (C.c_int, [_P, C.POINTER(Corr16Job), C.c_int, _I16]),
    "gsh_bank16_upload_jobs": (C.c_int, [_P, C.POINTER(Corr16Job), C.c_int]),
    "gsh_bank16_launch": (C.c_int, [_P]),
    "gsh_bank16_read_outputs": (C.c_int, [_P, _I16, C.c_int]),
    "gsh_bank16_time_launches": (C.c_int, [_P, C.c_int, _F]),
    "gsh_bank_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gsh_bank_destroy": (None, [_P]),
    "gsh_bank_set_code": (C.c_int, [_P, C.c_int, _F, C.c_int]),
    "gsh_bank_set_stream_host": (C.c_int, [_P, _F, C.c_uint64]),
    "gsh_bank_set_stream_device": (C.c_int, [_P, _P, C.c_uint64]),
    "gsh_bank_correlate": (C.c_int, [_P, C.POINTER(CorrJob), C.c_int, _F]),
    "gsh_bank_upload_jobs": (C.c_int, [_P, C.POINTER(CorrJob), C.c_int]),
    "gsh_bank_launch": (C.c_int, [_P, _P]),
    "gsh_bank_synchronize": (C.c_int, [_P]),
    "gsh_bank_read_outputs": (C.c_int, [_P, _F, C.c_int]),
    "gsh_bank_time_launches": (C.c_int, [_P, C.c_int, _F]),
    "gsh_bank_set_pair_fusion": (C.c_int, [_P, C.c_int]),
    "gsh_bank_set_splits": (C.c_int, [_P, C.c_int]),
    "gsh_bank_set_sample_base": (C.c_int, [_P, C.c_uint64]),
    "gsh_bank_set_stream_ring": (C.c_int, [_P, _P]),
    "gsh_stream_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "gsh_stream_destroy": (None, [_P]),
    "gsh_stream_push": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_push_device": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, _P, C.POINTER(C.c_uint64)]),
    "gsh_stream_push_async": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_wait": (C.c_int, [_P]),
    "gsh_stream_push_staged": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_push_pinned": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_push_pinned_async": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_wait_copied": (C.c_int, [_P]),
    "gsh_stream_wait_copied_upto": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gsh_host_register": (C.c_int, [C.c_int, _P, C.c_size_t]),
    "gsh_host_unregister": (C.c_int, [_P]),
    "gsh_stream_seek": (C.c_int, [_P, C.c_uint64]),
    "gsh_comm_unique_id": (C.c_int, [_P]),
    "gsh_stream_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "gsh_stream_group_create_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "gsh_stream_group_destroy": (None, [_P]),
    "gsh_stream_group_size": (C.c_int, [_P]),
    "gsh_stream_group_ring": (_P, [_P, C.c_int]),
    "gsh_stream_group_push": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_group_push_device": (C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "gsh_stream_group_wait": (C.c_int, [_P]),
    "gsh_stream_group_rccl_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "gsh_stream_group_plan": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "gsh_comm_library": (C.c_int, [C.c_char_p, C.c_int]),
    "gsh_stream_range": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gsh_stream_read": (C.c_int, [_P, C.c_uint64, C.c_uint64, _F]),
    "gsh_convert_samples_device": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, C.c_uint64, _P]),
    "gsh_fir_create": (C.c_int, [C.c_int, _F, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(_P)]),
    "gsh_fir_destroy": (None, [_P]),
    "gsh_fir_process_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    "gsh_direct_resample_device": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_uint64, _P, C.c_uint64,
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P]),
    "gsh_trk_create": (C.c_int, [C.c_int, C.POINTER(TrkConf), C.c_int, C.c_int, C.POINTER(_P)]),
    "gsh_trk_destroy": (None, [_P]),
    "gsh_trk_set_stream_host": (C.c_int, [_P, _F, C.c_uint64]),
    "gsh_trk_set_stream_device": (C.c_int, [_P, _P, C.c_uint64]),
    "gsh_trk_set_stream_ring": (C.c_int, [_P, _P]),
    "gsh_trk_start": (C.c_int, [_P, C.c_int, _F, _F, C.c_int, C.c_uint64, C.c_uint64, C.c_double]),
    "gsh_trk_start_ex": (C.c_int, [_P, C.c_int, _F, _F, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double]),
    "gsh_trk_pull_in_over": (C.c_int, [C.POINTER(TrkConf), C.c_uint64, C.c_uint64]),
    "gsh_trk_start_flags": (C.c_int, [_P, C.c_int, _F, _F, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_uint32]),
    "gsh_trk_pull_in": (C.c_int, [C.POINTER(TrkConf), C.c_uint64, C.c_double, C.c_uint64, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "gsh_trk_stop": (C.c_int, [_P, C.c_int]),
    "gsh_trk_run": (C.c_int, [_P, C.c_int, C.POINTER(TrkEpoch), C.POINTER(C.c_int32)]),
    "gsh_trk_set_split": (C.c_int, [_P, C.c_int]),
    "gsh_trk_run_begin": (C.c_int, [_P, C.c_int, C.c_int]),
    "gsh_trk_run_end": (C.c_int, [_P, C.POINTER(TrkEpoch), C.POINTER(C.c_int32)]),
    "gsh_trk_positions": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "gsh_trk_live_configure": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "gsh_trk_live_begin": (C.c_int, [_P]),
    "gsh_trk_live_in_flight": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "gsh_trk_live_take": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_int, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gsh_trk_live_quiesce": (C.c_int, [_P]),
    "gsh_trk_time_run": (C.c_int, [_P, C.c_int, C.c_int, _F]),
    "gsh_trk_write_dump": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(TrkConf), C.c_uint32, C.POINTER(TrkEpoch), C.c_int, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint32)]),
    "gsh_acq_create": (C.c_int, [C.c_int, C.POINTER(AcqConf), C.POINTER(_P)]),
    "gsh_acq_destroy": (None, [_P]),
    "gsh_acq_set_local_code": (C.c_int, [_P, C.c_uint32, _F]),
    "gsh_spectrum_peak": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "gsh_pb_create": (C.c_int, [C.c_int, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "gsh_pb_destroy": (None, [_P]),
    "gsh_pb_threshold": (C.c_float, [_P]),
    "gsh_pb_process_device": (C.c_int, [_P, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]),
    "gsh_pb_get_state": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gsh_notch_create": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "gsh_notch_destroy": (None, [_P]),
    "gsh_notch_threshold": (C.c_float, [_P]),
    "gsh_notch_process_device": (C.c_int, [_P, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]),
    "gsh_notch_get_state": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "gsh_acq_set_doppler_center": (C.c_int, [_P, C.c_int32]),
    "gsh_acq_set_doppler_bias": (C.c_int, [_P, C.c_int32]),
    "gsh_acq_set_grid_weight": (C.c_int, [_P, C.c_float]),
    "gsh_acq_input_power": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "gsh_acq_stage_input": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "gsh_acq_stage_input_device": (C.c_int, [_P, C.c_void_p]),
    "gsh_acq_dwell_resident": (C.c_int, [_P, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_time_correlate": (C.c_int, [_P, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_float)]),
    "gsh_acq_read_row_peaks": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "gsh_acq_noncoherent_pair_peaks": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gsh_acq_dwell": (C.c_int, [_P, _F, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_dwell_device": (C.c_int, [_P, _P, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_dwell_slots": (C.c_int, [_P, _F, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(AcqResult)]),
    "gsh_acq_dwell_step2": (C.c_int, [_P, _F, C.c_uint32, C.POINTER(C.c_uint32), _F, _F, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_dwell_step2_device": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32), _F, _F, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_dwell_cshort": (C.c_int, [_P, C.POINTER(C.c_int16), C.c_uint32, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_dwell_ring": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(AcqResult)]),
    "gsh_acq_read_grid": (C.c_int, [_P, C.c_uint32, _F]),
    "gsh_acq_time_dwells": (C.c_int, [_P, C.c_uint32, C.c_int, _F]),
    "gsh_acq_time_dwells_pipelined": (C.c_int, [_P, C.c_uint32, C.c_int, _F]),
    "gsh_acq_compute_threshold": (C.c_float, [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]),
}

_lib = None


def load():
    """Load libgnss_sdr_hip.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP engine has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name, None)
            if fn is None:
                continue  # reported by missing_symbols(); calling it later raises AttributeError
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def missing_symbols() -> list:
    """Names declared in include/gnss_sdr_hip.h (SYMBOLS) that the built library does not export."""
    lib = load()
    return [name for name in SYMBOLS if getattr(lib, name, None) is None]


def check(rc: int) -> None:
    if rc != GSH_OK:
        raise GshError(rc, load().gsh_last_error().decode("utf-8", "replace"))


def fptr(arr):
    """float32/complex64 numpy array -> float*."""
    return arr.ctypes.data_as(_F)
