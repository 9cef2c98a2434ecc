"""The tracking dump writer (gsh_trk_write_dump): an independent numpy dtype of log_data's layout (trk.cc:1599-1702, 108 bytes per record) is used.  No GPU needed: the records come from the CPU oracle loop (same struct layout as gsh_trk_epoch)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from helpers import synth_gps_l1_stream

LOG_DATA = np.dtype([("VE", "<f4"), ("E", "<f4"), ("P", "<f4"), ("L", "<f4"), ("VL", "<f4"), ("prompt_I", "<f4"), ("prompt_Q", "<f4"),
                     ("PRN_start_sample", "<u8"), ("acc_carrier_phase_rad", "<f4"), ("carrier_doppler_hz", "<f4"),
                     ("carrier_doppler_rate_hz_s", "<f4"), ("code_freq_hz", "<f4"), ("code_freq_rate_hz_s", "<f4"), ("carr_error", "<f4"),
                     ("carr_nco", "<f4"), ("code_error", "<f4"), ("code_nco", "<f4"), ("CN0_SNV_dB_Hz", "<f4"), ("carrier_lock_test", "<f4"),
                     ("var1", "<f4"), ("var2", "<f8"), ("PRN", "<u4"), ("TOW_ms", "<u8"), ("WN", "<u4")])


def _records():
    fs, n, epochs = 2.046e6, 2046, 120
    x = synth_gps_l1_stream((epochs + 3) * n, fs, [4], [1500.0], [0.0], cn0_dbhz=47.0, seed_noise=5)
    conf = oracle.trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=25.0, dll_bw_hz=2.0, enable_lock_detectors=1)
    return conf, oracle.trk_run(conf, oracle.ca_code(4), x, 0, 0, 1490.0, epochs)


def test_dump_layout_and_values(gsh, tmp_path):
    from gnss_sdr_amd._lib import TrkConf, TrkEpoch
    from gnss_sdr_amd.tracking_loop import write_dump
    assert LOG_DATA.itemsize == 108   # trk.cc:1705-1710 (save_matfile's epoch_size_bytes)
    conf_o, rec_o = _records()
    conf = TrkConf.from_buffer_copy(bytes(memoryview(conf_o)))          # same layout by construction (oracle/gnss_oracle.h)
    rec = [TrkEpoch.from_buffer_copy(bytes(memoryview(r))) for r in rec_o]
    path = tmp_path / "trk_dump_ch0.dat"
    write_dump(path, conf, 4, rec[:70])
    write_dump(path, conf, 4, rec[70:], append=True)
    d = np.fromfile(path, dtype=LOG_DATA)
    assert len(d) == len(rec)
    for k, r in enumerate(rec):
        assert d["P"][k] == np.float32(np.hypot(np.float32(r.corr[2]), np.float32(r.corr[3])))
        assert d["E"][k] == np.float32(np.hypot(np.float32(r.corr[0]), np.float32(r.corr[1])))
        assert d["VE"][k] == 0.0 and d["VL"][k] == 0.0
        assert d["prompt_I"][k] == np.float32(r.corr[2]) and d["prompt_Q"][k] == np.float32(r.corr[3])
        assert d["PRN_start_sample"][k] == r.sample_counter + r.prn_length_samples == int(d["var2"][k])
        assert d["carrier_doppler_hz"][k] == np.float32(r.carrier_doppler_hz)
        assert d["code_freq_hz"][k] == np.float32(r.code_freq_chips)
        assert d["carr_error"][k] == np.float32(r.carr_phase_error_hz) and d["carr_nco"][k] == np.float32(r.carr_error_filt_hz)
        assert d["code_error"][k] == np.float32(r.code_error_chips) and d["code_nco"][k] == np.float32(r.code_error_filt_chips)
        assert d["CN0_SNV_dB_Hz"][k] == np.float32(r.cn0_db_hz) and d["carrier_lock_test"][k] == np.float32(r.carrier_lock_test)
        assert d["var1"][k] == np.float32(r.rem_code_phase_samples) and d["PRN"][k] == 4
    # TOW hand-back columns (trk.cc:1921-1935 -> :1690-1695): one value per record, zeros when the caller has none
    assert not d["TOW_ms"].any() and not d["WN"].any()
    tow = [518400000 + 20 * k for k in range(len(rec))]
    write_dump(path, conf, 4, rec, tow_ms=tow, wn=[2200] * len(rec))
    d = np.fromfile(path, dtype=LOG_DATA)
    assert len(d) == len(rec) and np.array_equal(d["TOW_ms"], np.array(tow, dtype=np.uint64)) and np.all(d["WN"] == 2200)
    # (utils/python/lib/dll_pll_veml_read_tracking_dump.py predates the TOW / WN fields of the block's record and mis-steps through a current
    # dump file; the layout authority is the block itself: tests/host/test_tracking_adapters compares this writer's file with the file the
    # reference block writes with dump=true, record for record.)
