"""PIN of oracle/pcps_oracle.py to the reference itself (CPU).

oracle/_ref/libgnsssdr_ref_acq.so is the reference's own pcps_acquisition.cc (and the Tong / Galileo 8 ms / CCCWSR / QuickSync /
fine-Doppler blocks) compiled from /root/reference against stand-ins for GNU Radio, VOLK and FFTW (oracle/ref_acq_api.cc,
oracle/Makefile).  The blocks are driven through general_work the way the scheduler drives them.

Two transforms sit behind the blocks' gr::fft objects:
  * "pocketfft32" -- scipy's single-precision transform, the one the restatement uses.  Everything else being the reference's own
    code, block and restatement must then agree VALUE FOR VALUE: wipe-off tables, conjugated code spectra, every cell of the
    magnitude grid, peak indices, input power, statistic, threshold, Gnss_Synchro fields, state and "events" messages.
  * "double" -- the exact-definition DFT evaluated in float64 (oracle/ref_fft.cc).  Peak indices, Doppler, delay, state and events
    must not depend on the transform; grid values agree to float32 FFT rounding.
"""
import numpy as np
import pytest

import oracle
from oracle import ref_acq
from oracle.pcps_oracle import (CccwsrOracle, E5aNoncoherentIqOracle, FineDopplerOracle, Galileo8msOracle, PcpsOracle, QuickSyncOracle, TongOracle, compute_threshold,
                                count_doppler_bins)
from detector_cases import cccwsr_case, e1_8ms_case, e5a_case, fine_doppler_case, quicksync_case, tong_case
from helpers import synth_gps_l1_stream

pytestmark = pytest.mark.skipif(not ref_acq.available(), reason="oracle/_ref/libgnsssdr_ref_acq.so not built (needs /root/reference at build time)")

GPS_CHIP_RATE = 1.023e6
GPS_OPT_FS = 2000000.0   # GPS_L1_CA_OPT_ACQ_FS_SPS (GPS_L1_CA.h)


@pytest.fixture(params=["pocketfft32", "double"])
def transform(request):
    ref_acq.set_fft(request.param)
    yield request.param
    ref_acq.set_fft("double")


def gps_block(fs, role="Acquisition_1C", prn=1, signal="1C", system="G", **props):
    p = {"GNSS-SDR.internal_fs_sps": fs, role + ".blocking": "true"}
    for k, v in props.items():
        p[(role + "." + k) if not k.startswith("GNSS-SDR") else k] = v
    return ref_acq.RefAcqBlock(ref_acq.K_PCPS, p, GPS_CHIP_RATE, GPS_OPT_FS, 1, role=role, prn=prn, signal=signal, system=system)


def feed_one_dwell(b, x, chunk):
    """general_work calls until the block has run one acquisition_core (its state returns to 0 or 1 with the buffer consumed)."""
    pos, calls = 0, 0
    st0 = b.status()
    while True:
        avail = min(chunk, len(x) - pos)
        assert avail > 0, "stream exhausted before the dwell ran"
        state_before = b.status()["state"]
        _, c = b.work(x[pos:pos + avail])
        pos += c
        calls += 1
        if state_before == 2:
            return pos, calls
        assert calls < 1000


def assert_grid(ref_grid, ora_grid, transform):
    if transform == "pocketfft32":
        assert np.array_equal(ref_grid, ora_grid)
    else:
        assert np.max(np.abs(ref_grid - ora_grid)) <= 2e-6 * np.max(ora_grid)   # float32 FFT rounding of the restatement


CASES = [
    # fs, doppler_max, step, extra props, description
    dict(fs=4000000, dmax=5000, dstep=250, props=dict(pfa=0.01), prn=1, dop=1680.0, chips=200.5),
    dict(fs=4000000, dmax=5000, dstep=250, props=dict(threshold=2.5), prn=7, dop=-2300.0, chips=1000.2),           # peak-ratio statistic
    dict(fs=2046000, dmax=10000, dstep=500, props=dict(pfa=0.001), prn=22, dop=4210.0, chips=3.2),
    dict(fs=4000000, dmax=5000, dstep=250, props=dict(pfa=0.01, bit_transition_flag="true"), prn=3, dop=800.0, chips=511.0),
    dict(fs=4000000, dmax=5000, dstep=250, props=dict(threshold=2.0, bit_transition_flag="true"), prn=3, dop=800.0, chips=1020.9),
    dict(fs=2000000, dmax=5000, dstep=125, props=dict(pfa=0.01, coherent_integration_time_ms=2), prn=15, dop=-440.0, chips=77.7),  # fft = 2*consumed
    dict(fs=25000000, dmax=5000, dstep=250, props=dict(pfa=0.01), prn=5, dop=3120.0, chips=640.25),                 # BASELINE config 3 shape (one PRN)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"fs{c['fs']}_" + "_".join(f"{k}{v}" for k, v in c["props"].items()))
def test_single_dwell_matches_reference_block(case, transform):
    fs, prn = case["fs"], case["prn"]
    if fs == 25000000 and transform == "double":
        pytest.skip("covered by the pocketfft32 run; the float64 DFT of 82 x 25 000 points is slow")
    spms = fs // 1000
    props = dict(doppler_max=case["dmax"], doppler_step=case["dstep"], **case["props"])
    b = gps_block(fs, prn=prn, **props)
    st = b.status()
    consumed, fft_size, eff, nb = st["consumed_samples"], st["fft_size"], st["effective_fft_size"], st["num_doppler_bins"]
    sampled_ms = int(case["props"].get("coherent_integration_time_ms", 1))
    bt = case["props"].get("bit_transition_flag") == "true"
    # sizes (acq.cc:110-113)
    assert consumed == sampled_ms * spms * (2 if bt else 1)
    assert fft_size == (consumed if sampled_ms == 1 else 2 * consumed) and eff == (fft_size // 2 if bt else fft_size)
    x = synth_gps_l1_stream(consumed + 3000, fs, [prn], [case["dop"]], [case["chips"]], cn0_dbhz=49.0, seed_noise=fs % 977 + prn)
    # the adapter's local code: sampled_ms (x2 for bit transition) periods of the sampled replica (base_pcps_acquisition.cc code_ / vector_length_)
    one = oracle.ca_code_complex_sampled(prn, fs)
    code = np.tile(one, consumed // spms)
    b.set_local_code(code)
    b.set_active(True)
    pos, calls = feed_one_dwell(b, x, chunk=max(1000, spms // 3 + 17))
    st = b.status()
    use_cfar = "pfa" in case["props"]
    o = PcpsOracle(fs_in=fs, fft_size=fft_size, doppler_max=case["dmax"], doppler_step=case["dstep"], samples_per_chip=st["conf_samples_per_chip"],
                   samples_per_code=st["conf_samples_per_code"], consumed_samples=consumed, bit_transition_flag=bt, use_cfar=use_cfar)
    assert o.n_bins == nb and o.effective == eff
    o.set_local_code(code)
    r = o.dwell(x[:consumed])
    # tables and spectra
    for d in (0, nb // 3, nb - 1):
        assert np.array_equal(b.wipeoff(d), o.wipe[d])
    if transform == "pocketfft32":
        assert np.array_equal(b.fft_codes(), o.fft_codes)
    assert_grid(b.grid(), o.grid, transform)
    # what the block publishes
    assert st["acq_doppler_hz"] == r["doppler_hz"]
    assert st["acq_delay_samples"] == r["acq_delay_samples"]
    assert st["acq_samplestamp_samples"] == consumed and st["sample_counter"] == consumed and pos == consumed
    assert st["fs"] == fs
    if use_cfar:
        thr = compute_threshold(float(case["props"]["pfa"]), eff, nb, 1)
        assert st["threshold"] == thr
        if transform == "pocketfft32":
            assert st["input_power"] == np.float32(r["input_power"])
        else:
            assert st["input_power"] == pytest.approx(r["input_power"], rel=2e-6)
    else:
        assert st["threshold"] == np.float32(case["props"]["threshold"])
    # a detectable signal: positive acquisition, message 1 on "events", block back to state 0 / inactive
    assert st["events"] == [1] and st["state"] == 0 and st["active"] == 0
    assert r["test_statistics"] > st["threshold"]
    # truth (delay within half a chip, Doppler within a bin) -- the reference's own pass criteria (gps_l1_ca_pcps_acquisition_gsoc2013_test.cc:384-401)
    delay_chips = (1023.0 - case["chips"]) % 1023.0
    got_chips = (st["acq_delay_samples"] % spms) * 1023.0 / spms
    err = abs(got_chips - delay_chips)
    assert min(err, 1023.0 - err) < 0.5
    assert abs(st["acq_doppler_hz"] - case["dop"]) < case["dstep"]


def test_scheduler_chunking_does_not_change_the_dwell(transform):
    """Buffering (acq.cc:786-811): the dwell is over the first `consumed` samples however the scheduler slices them; the state-1 call
    that finds the buffer already full consumes nothing and moves to state 2 (the check precedes the increment, :804-807)."""
    fs, spms = 4000000, 4000
    x = synth_gps_l1_stream(3 * spms, fs, [9], [-1234.0], [333.3], cn0_dbhz=47.0, seed_noise=5)
    code = oracle.ca_code_complex_sampled(9, fs)
    results = []
    for chunk in (4000, 1000, 777, 16000):
        b = gps_block(fs, prn=9, doppler_max=5000, doppler_step=250, pfa=0.01)
        b.set_local_code(code)
        b.set_active(True)
        log = b.run_stream(x, chunk)
        st = log[-1]
        results.append((st["acq_delay_samples"], st["acq_doppler_hz"], st["acq_samplestamp_samples"], st["input_power"], tuple(st["events"])))
        states = [l["state"] for l in log]
        assert states[0] == 1 and states[-2:] == [2, 0], states       # 0 -> 1 without consuming, ..., full -> 2, core -> 0
        assert log[0]["pos"] == 0 and log[-1]["pos"] == spms and log[-2]["pos"] == spms
    assert len(set(results)) == 1, results


def test_noise_only_non_coherent_dwells_and_negative_message(transform):
    """max_dwells = 3 on a noise-only stream: three grids accumulated (acq.cc:545-553), input power divided by the dwell count (:430),
    state 1 between dwells, message 2 when the integration is done (:639-642); sample stamps advance by the consumed block."""
    fs, spms = 2046000, 2046
    rng = np.random.default_rng(42)
    x = (rng.standard_normal(4 * spms) + 1j * rng.standard_normal(4 * spms)).astype(np.complex64)
    code = oracle.ca_code_complex_sampled(12, fs)
    for use_cfar in (True, False):
        props = dict(doppler_max=4000, doppler_step=500, max_dwells=3)
        props.update(dict(pfa=1e-4) if use_cfar else dict(threshold=50.0))
        b = gps_block(fs, prn=12, **props)
        b.set_local_code(code)
        b.set_active(True)
        o = PcpsOracle(fs_in=fs, fft_size=spms, doppler_max=4000, doppler_step=500, samples_per_chip=2, samples_per_code=b.status()["conf_samples_per_code"],
                       use_cfar=use_cfar)
        o.set_local_code(code)
        pos = 0
        for dwell in (1, 2, 3):
            adv, _ = feed_one_dwell(b, x[pos:], chunk=900)
            r = o.dwell(x[pos:pos + spms], dwell)
            pos += adv
            st = b.status()
            assert_grid(b.grid(), o.grid, transform)
            assert st["acq_doppler_hz"] == r["doppler_hz"] and st["acq_delay_samples"] == r["acq_delay_samples"]
            assert st["acq_samplestamp_samples"] == pos
            if use_cfar:
                assert st["input_power"] == pytest.approx(r["input_power"], rel=2e-6)
            if dwell < 3:
                assert st["state"] == 1 and st["active"] == 1 and st["events"] == [] and st["dwell_count"] == dwell
            else:
                assert st["state"] == 0 and st["active"] == 0 and st["events"] == [2] and st["dwell_count"] == 0
        assert pos == 3 * spms


@pytest.mark.parametrize("use_cfar,nb2,step2", [(True, 4, 125.0), (False, 5, 62.5), (True, 8, 31.25)])
def test_two_step_state_machine_matches_reference_block(transform, use_cfar, nb2, step2):
    """make_two_steps (acq.cc:605-632): first positive -> step two around the found Doppler on the NEXT block with the narrow grid
    (:294-301), statistic against step one's input power (:428 not recomputed), Acq_doppler_step = doppler_step2 (:598-601)."""
    fs, spms, prn = 4000000, 4000, 19
    x = synth_gps_l1_stream(3 * spms, fs, [prn], [1437.0], [901.1], cn0_dbhz=50.0, seed_noise=nb2)
    code = oracle.ca_code_complex_sampled(prn, fs)
    props = dict(doppler_max=5000, doppler_step=250, make_two_steps="true", second_nbins=nb2, second_doppler_step=step2)
    props.update(dict(pfa=0.01) if use_cfar else dict(threshold=2.5))
    b = gps_block(fs, prn=prn, **props)
    b.set_local_code(code)
    b.set_active(True)
    o = PcpsOracle(fs_in=fs, fft_size=spms, doppler_max=5000, doppler_step=250, samples_per_chip=4, samples_per_code=b.status()["conf_samples_per_code"], use_cfar=use_cfar)
    o.set_local_code(code)
    adv1, _ = feed_one_dwell(b, x, 1500)
    r1 = o.dwell(x[:spms])
    st1 = b.status()
    assert st1["step_two"] == 1 and st1["active"] == 1 and st1["state"] == 0 and st1["events"] == [] and st1["dwell_count"] == 0
    assert st1["acq_doppler_hz"] == r1["doppler_hz"] and st1["doppler_center_step_two"] == np.float32(r1["doppler_hz"])
    adv2, _ = feed_one_dwell(b, x[adv1:], 1500)
    r2 = o.dwell_step2(x[adv1:adv1 + spms], float(r1["doppler_hz"]), nb2, step2, input_power_step_one=r1["input_power"])
    st2 = b.status()
    for d in range(nb2):
        w = b.wipeoff(d, step_two=True)
        phase_step = np.float32(6.283185307179586) * np.float32(r2["freqs"][d]) / np.float32(fs)
        exp = np.empty(2 * spms, np.float32)
        import ctypes as C
        ph = C.c_float(0.0)
        oracle.lib().oracle_sincos(exp, float(-phase_step), C.byref(ph), spms)
        assert np.array_equal(w, exp.view(np.complex64))
    assert st2["acq_doppler_hz"] == r2["doppler_hz"] and st2["acq_delay_samples"] == r2["acq_delay_samples"]
    assert st2["acq_doppler_step"] == int(step2) and st2["acq_samplestamp_samples"] == adv1 + adv2
    assert st2["events"] == [1] and st2["active"] == 0 and st2["step_two"] == 0
    if use_cfar:
        assert st2["input_power"] == st1["input_power"]
        assert st2["threshold_step_two"] == compute_threshold(0.01, spms, nb2, 1)


def test_cshort_input_and_doppler_center(transform):
    """item_type=cshort (acq.cc:653-656: volk_gnsssdr_16ic_convert_32fc then the same path) and set_doppler_center (:737-746)."""
    fs, spms, prn = 4000000, 4000, 27
    xf = synth_gps_l1_stream(2 * spms, fs, [prn], [7300.0], [12.5], cn0_dbhz=50.0, seed_noise=3)
    xs = np.empty((2 * spms, 2), np.int16)
    xs[:, 0] = np.clip(np.rint(xf.real * 300.0), -32768, 32767)
    xs[:, 1] = np.clip(np.rint(xf.imag * 300.0), -32768, 32767)
    code = oracle.ca_code_complex_sampled(prn, fs)
    b = gps_block(fs, prn=prn, doppler_max=2000, doppler_step=250, pfa=0.01, item_type="cshort")
    assert b.status()["conf_it_size"] == 4
    b.set_doppler_center(7000)
    b.set_local_code(code)
    b.set_active(True)
    adv, _ = feed_one_dwell(b, xs.view(np.uint32).reshape(-1), 1300)
    st = b.status()
    o = PcpsOracle(fs_in=fs, fft_size=spms, doppler_max=2000, doppler_step=250, samples_per_chip=4, samples_per_code=st["conf_samples_per_code"])
    o.set_doppler_center(7000)
    o.set_local_code(code)
    xc = (xs[:spms, 0].astype(np.float32) + 1j * xs[:spms, 1].astype(np.float32)).astype(np.complex64)
    r = o.dwell(xc)
    assert_grid(b.grid(), o.grid, transform)
    assert st["acq_doppler_hz"] == r["doppler_hz"] == 7250 and st["acq_delay_samples"] == r["acq_delay_samples"] and st["events"] == [1]


def test_glonass_fdma_bias(transform):
    """is_fdma (acq.cc:253-272): for "1G" the wipe-off tables carry DFRQ1_GLO * GLONASS_PRN[prn] on top of the grid Doppler."""
    fs, spms = 4000000, 4000
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(2 * spms) + 1j * rng.standard_normal(2 * spms)).astype(np.complex64)
    code = (rng.integers(0, 2, spms) * 2 - 1).astype(np.complex64)
    b = gps_block(fs, role="Acquisition_1G", prn=1, signal="1G", system="R", doppler_max=5000, doppler_step=250, pfa=0.01)
    b.set_local_code(code)       # set_local_code is where the block looks the channel up (acq.cc:221-224)
    # GLONASS_PRN.at(1) = 1 -> +562 500 Hz (GLONASS_L1_L2_CA.h: DFRQ1_GLO = 0.5625e6)
    o = PcpsOracle(fs_in=fs, fft_size=spms, doppler_max=5000, doppler_step=250, samples_per_chip=8, samples_per_code=4000.0, doppler_bias=562500)
    for d in (0, 17, 39):
        assert np.array_equal(b.wipeoff(d), o.wipe[d])
    o.set_local_code(code)
    b.set_active(True)
    feed_one_dwell(b, x, 4000)
    o.dwell(x[:spms])
    assert_grid(b.grid(), o.grid, transform)


def test_acq_conf_derived_fields_and_resampler_scaling(transform):
    """Acq_Conf (acq_conf.cc:29-124): float samples_per_ms, ceil samples_per_chip, the automatic resampler's decimation search; and
    update_synchro's resampler branch (acq.cc:586-591)."""
    b = gps_block(25000000, doppler_max=5000, doppler_step=250, pfa=0.01)
    st = b.status()
    assert st["conf_samples_per_ms"] == np.float32(25000000) * np.float32(0.001) and st["conf_samples_per_chip"] == 25
    assert st["consumed_samples"] == 25000 and st["num_doppler_bins"] == 40
    b = gps_block(25000000, doppler_max=5000, doppler_step=250, pfa=0.01, **{"GNSS-SDR.use_acquisition_resampler": "true"})
    st = b.status()
    # 25e6 / 2e6 = 12.5 -> decimation 12 does not divide 25e6; 10 does -> resampled 2.5 Msps (acq_conf.cc:100-110)
    assert st["conf_resampler_ratio"] == 10.0 and st["conf_resampled_fs"] == 2500000 and st["consumed_samples"] == 2500
    assert st["conf_samples_per_chip"] == 3
    fs2, prn = 2500000, 14
    x = synth_gps_l1_stream(3 * 2500, fs2, [prn], [910.0], [321.0], cn0_dbhz=50.0, seed_noise=6)
    code = oracle.ca_code_complex_sampled(prn, fs2)
    b.set_resampler_latency(37)
    b.set_local_code(code)
    b.set_active(True)
    feed_one_dwell(b, x, 1000)
    st = b.status()
    o = PcpsOracle(fs_in=fs2, fft_size=2500, doppler_max=5000, doppler_step=250, samples_per_chip=3, samples_per_code=st["conf_samples_per_code"])
    o.set_local_code(code)
    r = o.dwell(x[:2500])
    assert st["acq_delay_samples"] == r["acq_delay_samples"] * 10.0 - 37.0
    assert st["acq_samplestamp_samples"] == 25000 and st["fs"] == 2500000 and st["acq_doppler_hz"] == r["doppler_hz"]


# ---------------------------------------------------------------------------------------------------------------------------
# the other detector blocks
# ---------------------------------------------------------------------------------------------------------------------------
def _detector_block(kind, fs, props, chip_rate, ms_per_code, code_length_chips, extra=(0, 0, 0), override=None, **sat):
    role = "Acquisition"
    p = {"GNSS-SDR.internal_fs_sps": fs}
    p.update({role + "." + k: v for k, v in props.items()})
    sampled_ms = int(props.get("coherent_integration_time_ms", ms_per_code))
    code_length = int(round(fs / (chip_rate / code_length_chips)))                 # base_pcps_acquisition_custom.cc:78-80
    ov = dict(num_codes=sampled_ms // ms_per_code, code_length=code_length, vector_length=code_length * (sampled_ms // ms_per_code))
    ov.update(override or {})
    return ref_acq.RefAcqBlock(kind, p, chip_rate, 4e6, ms_per_code, role=role, extra=extra, override=ov, **sat)


def test_tong_block(transform):
    x, kw, code = tong_case()
    b = _detector_block(ref_acq.K_TONG, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], threshold=kw["threshold"]),
                        GPS_CHIP_RATE, 1, 1023.0, extra=(kw["tong_init_val"], kw["tong_max_val"], kw["tong_max_dwells"]), prn=10)
    o = TongOracle(**kw)
    assert b.status()["num_doppler_bins"] == o.n_bins == count_doppler_bins(10000, 250)
    b.set_local_code(code)
    o.set_local_code(code)
    b.set_active(True)
    n = kw["fft_size"]
    b.work(x[:n].reshape(1, n))            # state 0 -> 1 (pcps_tong_acquisition_cc.cc:162-184), nothing consumed
    assert b.status()["state"] == 1 and b.status()["consumed_last"] == 0
    k = 0
    while o.state == 1:
        blk = x[k * n:(k + 1) * n]
        b.work(blk.reshape(1, n))
        o.work(blk)
        st = b.status()
        assert (st["state"] if st["state"] != 0 else 2) in (1, 2, 3)
        assert st["dwell_count"] == o.dwell_count and st["tong_count"] == o.tong_count
        assert st["acq_delay_samples"] == o.result["acq_delay_samples"] and st["acq_doppler_hz"] == o.result["doppler_hz"]
        assert st["acq_samplestamp_samples"] == (k + 1) * n
        # volk_32f_accumulator_s32f: the stand-in sums the 4000 float terms sequentially, the restatement in float64 -- up to a few 1e-6
        # apart, which is the spread VOLK's own SIMD flavours have among themselves
        assert st["input_power"] == pytest.approx(float(o.input_power), rel=5e-6)
        assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=2e-5)
        g = b.grid()
        assert np.max(np.abs(g - o.grid)) <= 2e-5 * np.max(o.grid)
        assert st["state"] == o.state
        k += 1
    assert o.state == 2 and k == 7
    b.work(x[k * n:(k + 1) * n].reshape(1, n))   # state 2: publishes, goes inactive (:303-330)
    st = b.status()
    assert st["events"] == [1] and st["active"] == 0 and st["state"] == 0


def test_tong_block_noise_only(transform):
    x, kw, code = tong_case(signal=False, seed=5)
    kw = dict(kw, threshold=0.004)
    b = _detector_block(ref_acq.K_TONG, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], threshold=kw["threshold"]),
                        GPS_CHIP_RATE, 1, 1023.0, extra=(1, 8, 9), prn=10)
    o = TongOracle(**kw)
    b.set_local_code(code)
    o.set_local_code(code)
    b.set_active(True)
    n = 4000
    b.work(x[:n].reshape(1, n))
    b.work(x[:n].reshape(1, n))
    o.work(x[:n])
    st = b.status()
    assert st["state"] == o.state == 3 and st["tong_count"] == o.tong_count == 0
    b.work(x[:n].reshape(1, n))
    assert b.status()["events"] == [2]


@pytest.mark.parametrize("flip", [False, True])
def test_galileo_8ms_block(transform, flip):
    x, kw, code, delay = e1_8ms_case(flip)
    n = kw["fft_size"]
    b = _detector_block(ref_acq.K_8MS, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], coherent_integration_time_ms=8,
                                                        max_dwells=kw["max_dwells"]),
                        1.023e6, 4, 4092.0, override=dict(threshold=kw["threshold"]), system="E", signal="1B", prn=11)
    st = b.status()
    assert st["fft_size"] == n and int(st["conf_samples_per_code"]) == 16000   # float: 4000.00024 * 4
    o = Galileo8msOracle(**kw)
    b.set_local_code(code)
    o.set_local_code(code)
    b.set_active(True)
    b.work(x[:n].reshape(1, n))
    b.work(x[:n].reshape(1, n))
    o.work(x[:n])
    st = b.status()
    assert st["state"] == o.state == 2
    assert st["acq_doppler_hz"] == o.result["doppler_hz"] and st["acq_delay_samples"] == o.result["acq_delay_samples"]
    assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=2e-5)
    assert st["mag"] == pytest.approx(float(o.mag), rel=2e-5)
    b.work(x[:n].reshape(1, n))
    assert b.status()["events"] == [1]


@pytest.mark.parametrize("mode,ds,ps", [("inphase", 1.0, -1.0), ("quadrature", 1.0, 1.0), ("quadrature", 1.0, -1.0)])
def test_cccwsr_block(transform, mode, ds, ps):
    x, kw, cdata, cpilot, delay = cccwsr_case(mode, ds, ps)
    n = kw["fft_size"]
    b = _detector_block(ref_acq.K_CCCWSR, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], max_dwells=kw["max_dwells"]),
                        1.023e6, 4, 4092.0, override=dict(threshold=kw["threshold"]), system="E", signal="1B", prn=10)
    o = CccwsrOracle(**kw)
    b.set_local_code(cdata, cpilot)
    o.set_local_code(cdata, cpilot)
    b.set_active(True)
    b.work(x[:n].reshape(1, n))
    b.work(x[:n].reshape(1, n))
    o.work(x[:n])
    st = b.status()
    assert st["state"] == o.state == 2
    assert st["acq_doppler_hz"] == o.result["doppler_hz"] and st["acq_delay_samples"] == o.result["acq_delay_samples"]
    assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=2e-5)


@pytest.mark.parametrize("fs,p", [(8000000, 4), (4000000, 4), (8000000, 2)])
def test_quicksync_block(transform, fs, p):
    x, kw, code = quicksync_case(fs, p)
    spc = kw["samples_per_code"]
    b = _detector_block(ref_acq.K_QUICKSYNC, fs, dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], coherent_integration_time_ms=p),
                        GPS_CHIP_RATE, 1, 1023.0, extra=(p, kw["max_dwells"]), override=dict(threshold=kw["threshold"]), prn=10)
    o = QuickSyncOracle(**kw)
    st = b.status()
    assert st["fft_size"] == o.fft_size and st["num_doppler_bins"] == o.n_bins
    b.set_local_code(code)
    o.set_local_code(code)
    b.set_active(True)
    n = spc * p
    b.work(x[:n].reshape(1, n))
    b.work(x[:n].reshape(1, n))
    o.work(x[:n])
    st = b.status()
    assert st["state"] == o.state == 2
    assert st["acq_doppler_hz"] == o.result["doppler_hz"] and st["acq_delay_samples"] == o.result["acq_delay_samples"]
    assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=5e-5)


def test_fine_doppler_block(transform):
    """pcps_acquisition_fine_doppler_cc through all its states: 0 -> 1 (two 1 ms dwells accumulated) -> 2 (peak ratio, compute_CAF)
    -> 3 (collect 10 ms, estimate_Doppler on the eightfold zero-padded 320 000-point transform) -> 4 (message 1).  The block is the file as
    written: bin i wiped off at doppler_step*i - doppler_step (:170) but reported as i*doppler_step - doppler_max (:243)."""
    x, kw, code = fine_doppler_case()
    n = 4000
    b = _detector_block(ref_acq.K_FINE_DOPPLER, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], max_dwells=kw["max_dwells"],
                                                                  threshold=kw["threshold"]),
                        GPS_CHIP_RATE, 1, 1023.0, override=dict(samples_per_ms=float(n)), prn=10)     # gps_l1_ca_pcps_acquisition_fine_doppler.cc:52
    o = FineDopplerOracle(**kw)
    st = b.status()
    assert st["num_doppler_bins"] == o.n_points and st["fft_size"] == n
    b.set_local_code(code)
    o.set_local_code(code)
    b.set_active(True)
    pos = 0
    b.work(x[pos:pos + n], n)                        # state 0 -> 1
    assert b.status()["state"] == 1
    for k in range(2):
        _, c = b.work(x[pos:pos + n], n)
        assert c == n
        o.dwell(x[pos:pos + n])
        pos += c
        g = b.grid()
        tol = 0.0 if transform == "pocketfft32" else 2e-6 * np.max(o.p.grid)
        assert np.max(np.abs(g - o.p.grid)) <= tol
    assert b.status()["state"] == o.state == 2
    b.work(x[pos:pos + n], n)                        # state 2: decide
    o.decide()
    st = b.status()
    assert st["state"] == o.state == 3
    assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=1e-6)
    assert st["acq_delay_samples"] == o.result["acq_delay_samples"] and st["acq_doppler_hz"] == o.result["doppler_hz"] == -3000.0
    while b.status()["state"] == 3:                  # state 3: fills the 10 ms buffer from the stream, then estimates
        _, c = b.work(x[pos:pos + n], n)
        pos += c
    o.estimate_doppler(x[2 * n:])
    st = b.status()
    assert st["state"] == o.state == 4 and pos == 10 * n
    assert st["acq_doppler_hz"] == o.result["doppler_hz"]          # the fine estimate (1737.5 Hz) fails the block's own 1 kHz check against -3000
    b.work(x[pos:pos + n], n)
    st = b.status()
    assert st["events"] == [1] and st["active"] == 0 and st["positive_acq"] == 1
    # noise only -> state 5 -> message 2
    x, kw, code = fine_doppler_case(signal=False, seed=5)
    b = _detector_block(ref_acq.K_FINE_DOPPLER, kw["fs_in"], dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], max_dwells=kw["max_dwells"],
                                                                  threshold=kw["threshold"]), GPS_CHIP_RATE, 1, 1023.0, override=dict(samples_per_ms=float(n)), prn=10)
    b.set_local_code(code)
    b.set_active(True)
    for k in range(5):
        b.work(x[k * n:(k + 1) * n] if k else x[:n], n)
        if b.status()["events"]:
            break
    assert b.status()["events"] == [2]


E5A_CASES = {
    # name: (case kwargs, expected sampled_ms in the block)
    "cfg1_32Msps_1ms": dict(fs=32000000, sampled_ms=1, doppler=2800.0, delay_chips=4475.0, doppler_max=10000, cn0=50.0),
    "cfg2_12Msps_3ms": dict(fs=12000000, sampled_ms=3),
    "3ms_data_flip_first": dict(fs=12000000, sampled_ms=3, data_signs=(-1, 1, 1), pilot_signs=(1, 1, 1), delay_chips=10.0),
    "3ms_both_flip_first": dict(fs=8000000, sampled_ms=3, data_signs=(-1, 1, 1), pilot_signs=(-1, 1, 1), delay_chips=10.0, doppler=-1300.0),
    "2ms_pilot_flip": dict(fs=8000000, sampled_ms=2, data_signs=(1, 1), pilot_signs=(-1, 1), delay_chips=7.0, doppler=900.0),
    "3ms_data_only": dict(fs=8000000, sampled_ms=3, both=False, data_signs=(-1, 1, 1), delay_chips=10.0),
    "1ms_data_only": dict(fs=10240000, sampled_ms=1, both=False, cn0=50.0),
    "3ms_caf": dict(fs=8000000, sampled_ms=3, caf_window_hz=1500, doppler=-2100.0),
    "1ms_caf_data_only": dict(fs=10240000, sampled_ms=1, both=False, caf_window_hz=1000, cn0=50.0, doppler=4900.0),
    "zero_padding": dict(fs=8000000, sampled_ms=2, zero_padding=1, cn0=50.0),
    "noise_only_2_dwells": dict(fs=8000000, sampled_ms=2, signal=False, max_dwells=2, n_blocks=3),
}


@pytest.mark.parametrize("name", list(E5A_CASES))
def test_e5a_noncoherent_iq_block(transform, name):
    """galileo_e5a_noncoherentIQ_acquisition_caf_cc itself (compiled in place) against its restatement: same state, delay, Doppler (after the CAF filter when
    there is one), d_mag / statistic to float round-off, through the block's own buffering states (0 -> 1 -> 2 -> 3 | 4).  Items are single samples here
    (the block's input signature is sizeof(gr_complex), :58; it buffers a block itself)."""
    x, kw, ci, cq = e5a_case(**E5A_CASES[name])
    n, fs = kw["fft_size"], kw["fs_in"]
    both = kw["both_signal_components"]
    if name.startswith("noise"):
        # the adapter's threshold (ThresholdComputeDoppler) is the single-row exponential law; the maximum of a SUM of two |.|^2 rows over the A / B
        # choices sits well above it, so on noise the block as configured declares a false alarm (block and restatement agree on that, checked
        # once: state 3).  Tripled, the negative branch (state 4, message 2) is exercised.
        kw["threshold"] *= 3.0
    props = dict(doppler_max=kw["doppler_max"], doppler_step=kw["doppler_step"], coherent_integration_time_ms=kw["sampled_ms"], max_dwells=kw["max_dwells"])
    b = _detector_block(ref_acq.K_E5A_CAF, fs, props, 10.23e6, 1, 10230.0, extra=(1 if both else 0, kw["caf_window_hz"], kw["zero_padding"]),
                        override=dict(threshold=kw["threshold"]), system="E", signal="5X" if both else "5I", prn=11)
    st = b.status()
    assert st["fft_size"] == n and int(st["conf_samples_per_code"]) == kw["samples_per_code"]
    o = E5aNoncoherentIqOracle(**kw)
    assert st["num_doppler_bins"] == o.n_bins
    b.set_local_code(ci, cq)
    o.set_local_code(ci, cq)
    b.set_active(True)
    pos = 0
    for dwell in range(kw["max_dwells"]):
        blk = x[pos:pos + n]
        b.work(blk)                 # state 0 -> 1 (first dwell) or 1: fills the buffer
        if dwell == 0:
            assert b.status()["state"] == 1
            b.work(blk)             # state 1: n items buffered and consumed
        assert b.status()["consumed_last"] == n
        b.work(x[pos + n:pos + 2 * n])   # state 1 again: buffer full -> state 2, nothing consumed (:290-297)
        assert b.status()["state"] == 2 and b.status()["consumed_last"] == 0
        b.work(x[pos + n:pos + 2 * n])   # state 2: the search over the buffered block
        o.work(blk)
        st = b.status()
        assert st["state"] == o.state, (name, dwell, st["state"], o.state)
        assert st["dwell_count"] == o.well_count
        assert st["mag"] == pytest.approx(float(o.mag), rel=3e-5)
        assert st["input_power"] == pytest.approx(float(o.input_power), rel=2e-5)     # sequential float32 sum (generic VOLK loop) vs float64 sum
        assert st["test_statistics"] == pytest.approx(float(o.test_statistics), rel=3e-5)
        assert st["acq_doppler_hz"] == o.result["doppler_hz"] and st["acq_delay_samples"] == o.result["acq_delay_samples"], (name, st, o.result)
        pos += n
    assert o.state == (4 if name.startswith("noise") else 3)
    b.work(x[:n])
    assert b.status()["events"] == ([2] if name.startswith("noise") else [1])
    if name == "3ms_data_flip_first":
        assert o.rows[o.result["index_doppler"]][0] == "IB"      # the inverted first period is found by the B combination
    if name == "cfg2_12Msps_3ms":
        assert o.rows[o.result["index_doppler"]][0] == "IA"
        assert abs(o.result["acq_delay_samples"] - 1000.0 * 12e6 / 10.23e6) <= 1.0 and o.result["doppler_hz"] == 250.0
