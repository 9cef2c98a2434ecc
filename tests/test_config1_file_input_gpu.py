"""BASELINE config 1 end to end on the GPU: GPS L1 C/A, 1 channel, fs = 4 Msps, 1 ms coherent, 3-tap E/P/L, samples from a FILE.

The file is what File_Signal_Source reads (src/algorithms/signal_source/adapters/file_source_base.cc:345-373, 513): headerless
raw items, gr_complex = interleaved little-endian float32 I,Q -- and, second case, item_type ibyte (interleaved int8) going
through the data_type_adapter arithmetic on the device.  Configuration values are those of
conf/File_input/GPS/gnss-sdr_GPS_L1_gr_complex.conf (internal_fs_sps 4 000 000, doppler_max 10000, doppler_step 250, pfa 0.01,
pll_bw_hz 40, dll_bw_hz 4).  Flow, as a Channel drives it (channel_fsm.cc:190-213): the file is streamed into the device ring in
blocks -> PCPS acquisition on the first block that holds 1 ms -> hand-over (Acq_delay_samples, Acq_doppler_hz,
Acq_samplestamp_samples as the tracking pull-in uses them, trk.cc:1936-1973) -> 1000 code periods of the DLL/PLL loop.
Every stage is checked against the oracle run on the same file."""
import numpy as np
import pytest

import oracle
from oracle.pcps_oracle import PcpsOracle, compute_threshold
from helpers import synth_gps_l1_stream

pytestmark = pytest.mark.gpu

FS, N, EPOCHS = 4000000, 4000, 1000


def _read_file_in_blocks(path, item_type, block_items):
    """what File_Signal_Source + the flowgraph do: a stream of fixed-size item blocks"""
    dt = np.float32 if item_type == "gr_complex" else np.int8
    raw = np.fromfile(path, dtype=dt)
    per = 2
    for i in range(0, len(raw), block_items * per):
        yield raw[i:i + block_items * per]


@pytest.mark.parametrize("item_type", ["gr_complex", "ibyte"])
def test_file_input_acquisition_to_tracking(gpu, tmp_path, item_type):
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank, compute_threshold as gpu_threshold
    from gnss_sdr_amd.sample_stream import SampleStream
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    total = (EPOCHS + 30) * N
    doppler, code_phase = 2170.0, 388.6
    x = synth_gps_l1_stream(total, FS, [1], [doppler], [code_phase], cn0_dbhz=46.0, seed_noise=0x5EED0001)
    path = tmp_path / f"capture_{item_type}.dat"
    if item_type == "gr_complex":
        x.view(np.float32).tofile(path)
        xf = x
    else:
        q = np.clip(np.round(np.stack([x.real, x.imag], axis=1) * 24.0), -127, 127).astype(np.int8)
        q.tofile(path)
        xf = (q[:, 0].astype(np.float32) + 1j * q[:, 1].astype(np.float32)).astype(np.complex64)   # ibyte_to_complex: plain cast

    # ---- ingest: the file goes to the device ring block by block (20 ms blocks), converted there
    ring = SampleStream(total + 2, 2 * N, device=gpu)
    pushed = 0
    for blk in _read_file_in_blocks(path, item_type, 20 * N):
        first = ring.push(blk if item_type == "gr_complex" else blk.reshape(-1, 2), item_type)
        assert first == pushed
        pushed += len(blk) // 2
    assert pushed == total
    assert np.array_equal(ring.read(0, 3 * N), xf[:3 * N])

    # ---- acquisition on the first millisecond (doppler_max 10000, step 250 -> 80 bins; CFAR threshold from pfa = 0.01)
    kw = dict(fs_in=FS, fft_size=N, doppler_max=10000, doppler_step=250, samples_per_chip=4, samples_per_code=float(N))
    acq = PcpsAcquisitionBank(device=gpu, max_prn=1, **kw)
    code = oracle.ca_code_complex_sampled(1, FS)
    acq.set_local_code(0, code)
    res = acq.dwell_ring(ring, 0, 1)[0]                # straight from the device ring: the samples crossed PCIe once
    assert res == acq.dwell(ring.read(0, N), 1)[0]     # ... and equal the dwell over the same samples handed over from the host
    ora = PcpsOracle(**kw)
    ora.set_local_code(code)
    exp = ora.dwell(xf[:N])
    assert (res["index_time"], res["index_doppler"], res["doppler_hz"]) == (exp["index_time"], exp["index_doppler"], exp["doppler_hz"])
    thr = gpu_threshold(0.01, N, 80, 1)
    assert thr == pytest.approx(compute_threshold(0.01, N, 80, 1), rel=2e-6)
    assert res["test_statistics"] > thr
    assert abs(res["doppler_hz"] - doppler) <= 250
    acq.close()

    # ---- hand-over, as the pull-in state computes it (trk.cc:1936-1973, without the Doppler-induced code-rate correction terms that
    #      vanish for a zero-delay hand-over): first window starts at the next code boundary after the acquisition stamp
    acq_stamp = N                                     # sample counter at the end of the dwell (acq.cc:592-596)
    delay = res["acq_delay_samples"]
    start = acq_stamp + int(round(delay))             # code start inside the NEXT period
    conf_kw = dict(fs_in=float(FS), vector_length=N, pll_bw_hz=40.0, dll_bw_hz=4.0, early_late_space_chips=0.5)
    loop = TrackingLoop(trk_conf(**conf_kw), 1, 1023, device=gpu)
    loop.set_stream_ring(ring)                        # the loop reads the same resident samples (absolute sample indices)
    loop.start(0, oracle.ca_code(1), start, acq_stamp, float(res["doppler_hz"]))
    rec, done = loop.run(EPOCHS)
    ora_rec = oracle.trk_run(oracle.trk_conf(**conf_kw), oracle.ca_code(1), xf, start, acq_stamp, float(res["doppler_hz"]), EPOCHS)
    assert done[0] == EPOCHS == len(ora_rec)
    # locked: Doppler estimate on the true value, prompt carries the signal, early and late balanced
    tail = rec[0][-300:]
    assert abs(np.mean([r.carrier_doppler_hz for r in tail]) - doppler) < 1.0
    p = np.mean([np.hypot(r.corr[2], r.corr[3]) for r in tail])
    e = np.mean([np.hypot(r.corr[0], r.corr[1]) for r in tail])
    l = np.mean([np.hypot(r.corr[4], r.corr[5]) for r in tail])
    assert p > 1.6 * e and p > 1.6 * l and abs(e - l) < 0.1 * p
    # and the same trajectory as the oracle loop (window positions identical until the loops' last float bits diverge; Doppler close)
    same = sum(1 for g, o in zip(rec[0], ora_rec) if g.sample_counter == o.sample_counter)
    assert same >= 0.9 * EPOCHS, same
    assert abs(np.mean([r.carrier_doppler_hz for r in tail]) - np.mean([r.carrier_doppler_hz for r in ora_rec[-300:]])) < 0.2
    loop.close()
    ring.close()
