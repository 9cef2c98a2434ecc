"""GPU parity of the fine-Doppler second step (make_two_steps, pcps_acquisition.cc:294-301, 428-437, 522-560 with d_step_two,
605-624) and of cshort input (acq.cc:653-656) against oracle/pcps_oracle.py.

Bars as in test_acquisition_gpu.py: (index_time, index_doppler) and the integer Doppler bit-exact with the oracle; grid values
within 2e-3 of the peak; CFAR statistic = peak / step-one power to float32 rounding of the peak."""
import numpy as np
import pytest

import oracle
from oracle.pcps_oracle import PcpsOracle
from helpers import synth_gps_l1_stream

pytestmark = pytest.mark.gpu


def _bank(gpu, **kw):
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    return PcpsAcquisitionBank(device=gpu, **kw)


@pytest.mark.parametrize("fs,nb2,step2,use_cfar,path", [(4000000, 4, 125.0, True, 0), (4000000, 5, 62.5, False, 0),
                                                        (25000000, 4, 125.0, True, 0), (2046000, 4, 125.0, True, 0),
                                                        (4000000, 8, 31.25, True, 1), (25000000, 6, 50.0, False, 0)])
def test_step_two_matches_oracle(gpu, fs, nb2, step2, use_cfar, path):
    n = fs // 1000
    prns = [3, 11, 19]
    dops = [1437.0, -2210.0, 333.0]
    x = synth_gps_l1_stream(2 * n, fs, prns, dops, [100.2, 640.7, 901.1], cn0_dbhz=50.0, seed_noise=fs + nb2)
    spc = int(np.ceil(fs / 1.023e6))
    kw = dict(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, samples_per_chip=spc, samples_per_code=float(n), use_cfar=use_cfar)
    acq = _bank(gpu, max_prn=3, num_doppler_bins_step2=nb2, doppler_step2=step2, transform_path=path, **kw)
    oras = []
    for i, p in enumerate(prns):
        code = oracle.ca_code_complex_sampled(p, fs)
        acq.set_local_code(i, code)
        o = PcpsOracle(**kw)
        o.set_local_code(code)
        oras.append(o)
    # step one on the first block
    r1 = acq.dwell(x[:n], 3)
    e1 = [o.dwell(x[:n]) for o in oras]
    for a, b in zip(r1, e1):
        assert (a["index_time"], a["index_doppler"], a["doppler_hz"]) == (b["index_time"], b["index_doppler"], b["doppler_hz"])
    # step two on the NEXT block (acq.cc:609-624: d_state = 0, a new buffer is collected), slots in a shuffled order
    order = [2, 0, 1]
    centers = [float(r1[i]["doppler_hz"]) for i in order]
    powers = [r1[i]["input_power"] for i in order]
    r2 = acq.dwell_step2(x[n:2 * n], order, centers, powers if use_cfar else None)
    for k, i in enumerate(order):
        e2 = oras[i].dwell_step2(x[n:2 * n], centers[k], nb2, step2, input_power_step_one=e1[i]["input_power"])
        got = r2[k]
        assert (got["index_time"], got["index_doppler"]) == (e2["index_time"], e2["index_doppler"]), (i, got, e2)
        assert got["doppler_hz"] == e2["doppler_hz"]
        assert got["acq_delay_samples"] == e2["acq_delay_samples"]
        assert got["peak"] == pytest.approx(e2["peak"], rel=2e-3)
        assert got["test_statistics"] == pytest.approx(e2["test_statistics"], rel=2e-3)
        if use_cfar:
            assert got["input_power"] == r1[i]["input_power"]        # step one's value is carried, not recomputed (acq.cc:428)
        # the refinement does what it is for: the Doppler estimate moves to within one fine bin of the truth
        assert abs(got["doppler_hz"] - dops[i]) <= step2
        g = acq.read_grid(i)[:nb2]
        assert np.max(np.abs(g - e2["grid"])) <= 2e-3 * e2["peak"]


def test_step_two_needs_configuration(gpu):
    from gnss_sdr_amd import GshError
    fs, n = 4000000, 4000
    acq = _bank(gpu, fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, samples_per_chip=4, samples_per_code=4000.0, max_prn=1)
    acq.set_local_code(0, oracle.ca_code_complex_sampled(1, fs))
    with pytest.raises(GshError) as e:
        acq.dwell_step2(np.zeros(n, np.complex64), [0], [0.0], [1.0])
    assert e.value.code == 4  # GSH_ERR_STATE
    with pytest.raises(GshError):  # narrow grid wider than the wide one
        _bank(gpu, fs_in=fs, fft_size=n, doppler_max=500, doppler_step=500, samples_per_chip=4, samples_per_code=4000.0, num_doppler_bins_step2=4)


@pytest.mark.parametrize("fs", [4000000, 25000000])
def test_cshort_input_equals_converted_float_input(gpu, fs):
    """acq.cc:653-656: volk_gnsssdr_16ic_convert_32fc is an exact int16 -> float cast, so a cshort dwell must equal the
    gr_complex dwell over the converted samples bit for bit."""
    n = fs // 1000
    x = synth_gps_l1_stream(n, fs, [5, 9], [800.0, -3100.0], [12.0, 700.5], cn0_dbhz=48.0, seed_noise=77)
    x16 = np.round(np.stack([x.real, x.imag], axis=1) * 400.0).astype(np.int16)
    xf = (x16[:, 0].astype(np.float32) + 1j * x16[:, 1].astype(np.float32)).astype(np.complex64)
    kw = dict(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, samples_per_chip=int(np.ceil(fs / 1.023e6)), samples_per_code=float(n))
    acq = _bank(gpu, max_prn=2, **kw)
    for i, p in enumerate((5, 9)):
        acq.set_local_code(i, oracle.ca_code_complex_sampled(p, fs))
    a = acq.dwell_cshort(x16, 2)
    ga = [acq.read_grid(i) for i in range(2)]
    b = acq.dwell(xf, 2)
    gb = [acq.read_grid(i) for i in range(2)]
    assert a == b
    for u, v in zip(ga, gb):
        assert np.array_equal(u, v)
    o = PcpsOracle(**kw)
    o.set_local_code(oracle.ca_code_complex_sampled(5, fs))
    e = o.dwell(xf)
    assert (a[0]["index_time"], a[0]["index_doppler"]) == (e["index_time"], e["index_doppler"])
