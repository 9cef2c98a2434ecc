#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: tracking correlators/s.

Workload (config.workload): BASELINE config 2 -- GPS L1 C/A, 32 channels (PRN 1..32), fs = 25 Msps,
N = 25 000 samples per 1 ms epoch, 3-tap E/P/L -- open-loop (pre-computed NCO parameter table), every channel
reading its own window sequence of ONE shared complex64 IF stream (noise + 8 embedded signals at 45 dB-Hz).
A "step" is one pass of the hot path over one batch: channels x epochs jobs in one launch, inputs (stream, codes,
job table) already resident in HBM.  value = channels*taps*epochs / time, whole job.

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU; every rank tracks its own 32 channels of the same
  stream (weak scaling).  Every step the ingest rank re-distributes the stream block over RCCL in the front-end's
  8-bit format (scatter + all-gather across all xGMI links, double-buffered on the communicator's stream and
  overlapped with the previous block's correlation), every rank converts it to complex64 on its GPU and
  correlates; all of that IS inside the timed region.

One JSON line on stdout (rank 0).  Besides the contract keys it carries
  roofline      -- dominant kernel (mcorr_kernel<3,0,false>) vs the HBM roofline, algorithmic bytes 8N+8T per job,
                   duration from HIP events on the launch stream
  cpu_baseline  -- the reference's own Cpu_Multicorrelator_Real_Codes (oracle/_ref, x86 SIMD protokernels) timed on
                   this box's host cores over a bounded sample (falls back to the C port when _ref is absent)
  acquisition   -- secondary metric: PCPS dwells/s for BASELINE config 3 (32 PRN x 41 bins x 25 000)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s measured achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=400, help="1 ms epochs per channel per step")
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--taps", type=int, default=3)
    ap.add_argument("--settle-steps", type=int, default=800,
                    help="untimed steps run during set-up, before the W warm-up steps, so that the GPU clocks have settled: after an idle "
                         "period the first ~40 ms of work run up to 25 %% slower (profiles/ab/clock_ramp.py); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-acq", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config 4 / config 5 figures (profiles/run_profiles.sh: keeps the "
                    "per-kernel averages of the trace about the headline workload only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    return ap.parse_args()


def make_stream_torch(torch, dev, n_samples, fs, n_sig=8, cn0=45.0):
    """Synthetic IF stream on the device: N(0,1)+jN(0,1) + n_sig GPS C/A signals at cn0 dB-Hz."""
    from gnss_sdr_amd.codes import gps_l1_ca_code
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0002)
    x = torch.randn(n_samples, 2, generator=g, device=dev, dtype=torch.float32)
    x = torch.view_as_complex(x).contiguous()
    rng = np.random.default_rng(0x5EED0003)
    dop = rng.uniform(-5000, 5000, n_sig)
    cph = rng.uniform(0, 1023, n_sig)
    amp = float(np.sqrt(10.0 ** (cn0 / 10.0) * 2.0 / fs))
    t = torch.arange(n_samples, device=dev, dtype=torch.float64)
    for i in range(n_sig):
        code = torch.from_numpy(gps_l1_ca_code(i + 1)).to(dev)
        f_code = 1.023e6 * (1.0 + dop[i] / 1575.42e6)
        chip = torch.floor(t * (f_code / fs) + cph[i]).to(torch.int64) % 1023
        ph = (2.0 * np.pi * dop[i] / fs) * t
        ph = ph - 2.0 * np.pi * torch.floor(ph / (2.0 * np.pi))
        x += (amp * code[chip]).to(torch.complex64) * torch.polar(torch.ones_like(ph, dtype=torch.float32), ph.to(torch.float32))
        del chip, ph
    return x, dop, cph


def build_jobs(channels, epochs, n, fs, taps, dop, cph, rank):
    """Epoch-major, channel-minor job table (jobs that read the same samples are adjacent: they share an XCD L2)."""
    from gnss_sdr_amd.tracking import make_jobs

    def tracking_params_for(fs, doppler_hz, rng):
        """Per-channel NCO parameters as the tracking block would pass them (trk.cc:1237-1243)."""
        two_pi = 6.283185307179586
        return dict(rem_carr_phase_rad=float(np.float32(rng.uniform(0.0, two_pi))),
                    phase_step_rad=float(np.float32(two_pi * doppler_hz / fs)),
                    rem_code_phase_chips=float(np.float32(rng.uniform(0.0, 1.0))),
                    code_phase_step_chips=float(np.float32(1.023e6 * (1.0 + doppler_hz / 1575.42e6) / fs)))
    rng = np.random.default_rng(0x5EED0004 + rank)
    per_ch = []
    for c in range(channels):
        if c < len(dop) and rank == 0:
            fd = float(dop[c])
            f_code = 1.023e6 * (1 + fd / 1575.42e6)
            start = (1023.0 - cph[c]) / f_code * fs
            off = int(np.ceil(start))
            p = tracking_params_for(fs, fd, rng)
            p["rem_code_phase_chips"] = float(np.float32(-(off - start) * f_code / fs))
        else:
            off = int(rng.integers(0, n))
            p = tracking_params_for(fs, rng.uniform(-5000, 5000), rng)
        per_ch.append((off, p))
    if taps == 3:
        shifts = [-0.5, 0.0, 0.5]
    elif taps == 5:
        shifts = [-0.5, -0.15, 0.0, 0.15, 0.5]
    else:
        shifts = list(np.linspace(-0.5, 0.5, taps))
    rows = []
    for e in range(epochs):
        for c in range(channels):
            off, p = per_ch[c]
            rows.append(dict(sample_offset=off + e * n, n_samples=n, code_slot=c, shifts_chips=shifts, **p))
    return make_jobs(rows), rows


def cpu_baseline(channels, n, fs, taps, target_s):
    """The reference CPU path on this box's host cores over a bounded sample of the same workload."""
    import oracle
    from helpers import synth_gps_l1_stream, tracking_params_for
    cores = os.cpu_count() or 1
    R = oracle.ref()
    kind = "reference" if R is not None else "port"
    epochs_stream = 8
    x = synth_gps_l1_stream((epochs_stream + 2) * n, fs, [1, 2], [1000.0, -2500.0], [10.0, 500.0], seed_noise=3)
    xi = np.ascontiguousarray(x).view(np.float32)
    codes = np.concatenate([oracle.ca_code(c % 32 + 1) for c in range(channels)]).astype(np.float32)
    shifts = np.array([-0.5, 0.0, 0.5] if taps == 3 else np.linspace(-0.5, 0.5, taps), np.float32)
    rng = np.random.default_rng(9)
    params = np.zeros((channels, 6), np.float32)
    for c in range(channels):
        p = tracking_params_for(fs, rng.uniform(-5000, 5000), rng)
        params[c] = [p["rem_carr_phase_rad"], p["phase_step_rad"], p["rem_code_phase_chips"], p["code_phase_step_chips"], rng.integers(0, n), 0]
    out = np.zeros(channels * 2 * taps, np.float32)

    threads = cores

    def run(epochs):
        if R is not None:
            R.ref_set_flavour(1)  # x86 SIMD protokernels = what volk_gnsssdr dispatches to on this host
            try:
                return R.ref_mcorr_time(codes, 1023, shifts, taps, xi, len(x), n, channels, epochs, threads, params, out)
            finally:
                R.ref_set_flavour(0)
        return oracle.lib().oracle_mcorr_time(codes, 1023, shifts, taps, xi, len(x), n, channels, epochs, threads, params, out)

    # pick the thread count that serves the CPU best (more threads than memory channels can hurt this streaming kernel)
    best = (0.0, cores)
    for cand in sorted({min(cores, c) for c in (8, 16, 32, 64, 128, 256, cores)}):
        threads = cand
        e = max(16, 2 * cand * 16 // channels)
        run(e)
        rate = channels * e / max(run(e), 1e-9)
        if rate > best[0]:
            best = (rate, cand)
    threads = best[1]
    cores = threads
    epochs = 64
    t = run(epochs)  # calibration, then grow the sample until it fills about target_s of wall time
    while t < target_s / 3.0 and epochs < 200000:
        epochs = int(min(200000, max(epochs * 2, epochs * 0.9 * target_s / max(t, 1e-6))))
        t = run(epochs)
    simd = bool(R is not None and R.ref_simd_supported())
    return {
        "value": channels * taps * epochs / t,
        "unit": "correlators/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{channels} channels x {epochs} epochs of {n} samples, {taps} taps, {cores} threads, "
                  + ("Cpu_Multicorrelator_Real_Codes over volk_gnsssdr " + ("u_avx" if simd else "generic") + " protokernels (oracle/_ref)"
                     if kind == "reference" else "oracle/gnss_oracle.c scalar port"),
        "seconds": t,
    }


def acquisition_metric(torch, dev_index, x_block, fs):
    """Secondary metric: PCPS dwells/s, BASELINE config 3 (32 PRN x 41 Doppler bins x 25 000 samples)."""
    try:
        from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    except Exception as e:  # acquisition not built yet
        return {"error": f"acquisition unavailable: {e}"}
    from gnss_sdr_amd.codes import gps_l1_ca_code_sampled
    n = int(fs * 1e-3)
    acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41,
                              samples_per_chip=int(np.ceil(fs / 1.023e6)), samples_per_code=float(n), max_prn=32, device=dev_index,
                              keep_grid=False)  # max_dwells = 1, dump = false: the statistics are formed on chip
    for p in range(32):
        acq.set_local_code(p, gps_l1_ca_code_sampled(p + 1, int(fs)))
    acq.time_dwells(x_block, 32, reps=400, pipelined=True)        # ~60 ms untimed: the clocks settle (profiles/ab/clock_ramp.py)
    ms_serial = acq.time_dwells(x_block, 32, reps=20)             # one batch after the other on one stream: latency
    ms = acq.time_dwells(x_block, 32, reps=200, pipelined=True)   # batches alternating on two streams: throughput
    nbytes = 16.0 * n * 41 * (32 + 1)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("acquisition", {})
        if tj.get("n") == n and tj.get("n_prn") == 32 and tj.get("n_bins") == 41:
            traffic = tj.get("hbm_bytes_per_batch")
    except Exception:
        traffic = None
    res = {"metric": "acquisition dwells/s", "value": 32.0 / (ms * 1e-3), "unit": "dwells/s", "ms_per_batch": ms,
           "ms_per_batch_single_stream": ms_serial,
           "config": {"workload": "GPS L1 C/A PCPS, 32 PRN x 41 Doppler bins, N=25000, 1 dwell"},
           "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_batch": nbytes,
                        "kernels": "oc_forward_kernel + oc_cell_kernel<Plan<25,25,40>,false,false>"}}
    acq.close()
    try:
        res["cpu_baseline"] = acquisition_cpu_baseline(x_block.cpu().numpy(), fs, n)
    except Exception as e:
        res["cpu_baseline"] = {"error": str(e)}
    return res


def _acq_cpu_worker(args):
    """One process of the acquisition CPU baseline: `reps` dwells (1 PRN x D bins each); returns its busy time."""
    x, fs, n, prn, reps = args
    import oracle
    from oracle.pcps_oracle import PcpsOracle
    o = PcpsOracle(int(fs), n, 5000, 250, int(np.ceil(fs / 1.023e6)), float(n), num_doppler_bins=41)
    o.set_local_code(oracle.ca_code_complex_sampled(prn, int(fs)))
    t0 = time.perf_counter()
    for _ in range(reps):
        o.dwell(x)
    return time.perf_counter() - t0


def acquisition_cpu_baseline(x, fs, n, target_s=6.0):
    """the CPU restatement of doppler_grid + statistics (oracle/pcps_oracle.py, float32, scipy pocketfft
    standing in for the reference's FFTW -- kind "port"), one dwell per PRN, PRNs spread over host processes."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = min(os.cpu_count() or 1, 64)
    _acq_cpu_worker((x, fs, n, 1, 1))                     # warm-up (pocketfft plan cache, page faults)
    t1 = _acq_cpu_worker((x, fs, n, 2, 2)) / 2.0          # single-process dwell time
    reps = int(max(1, min(64, target_s / max(t1, 1e-3))))
    with ProcessPoolExecutor(cores, mp_context=mp.get_context("spawn")) as ex:
        list(ex.map(_acq_cpu_worker, [(x, fs, n, (i % 32) + 1, 1) for i in range(cores)]))     # start + warm the workers
        t0 = time.perf_counter()
        list(ex.map(_acq_cpu_worker, [(x, fs, n, (i % 32) + 1, reps) for i in range(cores)]))
        dt = time.perf_counter() - t0
    n_dwells = cores * reps
    return {"value": n_dwells / dt, "unit": "dwells/s", "cores": cores, "kind": "port",
            "sample": f"{n_dwells} dwells (1 PRN x 41 bins x {n} samples each) over {cores} processes, numpy/scipy-pocketfft "
                      f"restatement of pcps_acquisition (the reference's FFTW/VOLK are not available here)",
            "single_process_dwells_per_s": 1.0 / t1, "seconds": dt}


def pcie_inclusive_metric(torch, dev_index, x_dev, jobs, C, E, T, n_samples, reps=5):
    """Not `value`: the same step when the stream block arrives from HOST memory -- 8-bit items (what a front-end delivers) in a pinned
    buffer -> gsh_stream_push (H2D + cast on the device) -> one launch over all channels / epochs, nothing overlapped."""
    from gnss_sdr_amd.sample_stream import SampleStream
    from gnss_sdr_amd.tracking import CorrelatorBank
    from gnss_sdr_amd.codes import gps_l1_ca_code
    q = torch.view_as_real(x_dev).mul(30.0).round_().clamp_(-127, 127).to(torch.int8).cpu()
    host = torch.empty_like(q).pin_memory()
    host.copy_(q)
    ring = SampleStream(n_samples + 2, n_samples // 2, device=dev_index)
    bank = CorrelatorBank(C, 1023, device=dev_index)
    for c in range(C):
        bank.set_code(c, gps_l1_ca_code(c % 32 + 1))
    bank.set_stream_ring(ring)
    bank.set_splits(1)
    h = host.numpy()
    from gnss_sdr_amd._lib import CorrJob
    base = np.frombuffer(jobs, dtype=np.dtype(CorrJob)).copy()
    tables = []
    for r in range(reps + 1):  # absolute sample indices: block r starts at r * n_samples (built outside the timed region, like the bench's table)
        t = base.copy()
        t["sample_offset"] += np.uint64(r * n_samples)
        tables.append((t, (CorrJob * len(t)).from_buffer(t)))
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        first = ring.push(h, "ibyte")
        assert first == r * n_samples
        bank.upload_jobs(tables[r][1])
        bank.launch()
        bank.synchronize()
        ts.append(time.perf_counter() - t0)
    bank.close()
    ring.close()
    dt = float(np.median(ts[1:]))
    return {"value": C * T * E / dt, "unit": "correlators/s", "ms_per_step": dt * 1e3, "host_bytes_per_step": int(h.nbytes),
            "note": "8-bit stream block from pinned host memory -> device ring (H2D + cast) -> job table upload -> launch; sequential, nothing overlapped"}


def closed_loop_metric(dev_index, x_dev, n_samples, fs, n, dop, cph, channels=32, epochs=200):
    """Secondary metric: the DLL/PLL loop closed on the device (gsh_trk_*), BASELINE config 2 shape -- every channel runs
    `epochs` consecutive code periods with its own discriminators / loop filters / NCO update between them, one launch."""
    try:
        from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    except Exception as e:
        return {"error": f"tracking loop unavailable: {e}"}
    from gnss_sdr_amd.codes import gps_l1_ca_code
    conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0)
    loop = TrackingLoop(conf, channels, 1023, device=dev_index)
    loop.set_stream_device(x_dev.data_ptr(), n_samples, keepalive=x_dev)
    rng = np.random.default_rng(0x5EED0006)
    for c in range(channels):
        if c < len(dop):  # hand-over from a (simulated) acquisition: code start of the embedded signal, Doppler off by <= 20 Hz
            f_code = 1.023e6 * (1 + dop[c] / 1575.42e6)
            start = int(round((1023.0 - cph[c]) / f_code * fs))
            loop.start(c, gps_l1_ca_code(c + 1), start, 0, float(dop[c]) + rng.uniform(-20, 20))
        else:
            loop.start(c, gps_l1_ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
    loop.time_run(epochs, reps=20)   # ~45 ms untimed: the clocks settle (profiles/ab/clock_ramp.py)
    ms = loop.time_run(epochs, reps=5)
    rec, done = loop.run(epochs)
    locked = 0
    for c in range(min(channels, len(dop))):
        tail = rec[c][-20:]
        if abs(np.mean([r.carrier_doppler_hz for r in tail]) - dop[c]) < 5.0:
            locked += 1
    loop.close()
    return {"metric": "correlators/s, loop closed on device", "value": channels * 3 * epochs / (ms * 1e-3), "unit": "correlators/s",
            "ms_per_launch": ms, "us_per_epoch": ms * 1e3 / epochs, "channels": channels, "epochs_per_launch": epochs,
            "channels_with_signal_locked": f"{locked}/{min(channels, len(dop))}",
            "real_time_factor": epochs * 1e-3 / (ms * 1e-3)}


def other_configs_metric(dev_index):
    """Secondary figures: the correlator bank at the shapes of BASELINE configs 4 and 5 (one GPU's 32 of the 256 channels), job tables
    from profiles/config_rates.py (random +-1 codes of the right lengths: the kernel's cost does not depend on the code values)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("config_rates", os.path.join(ROOT, "profiles", "config_rates.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for key, build in (("config4_galileo_e1_50ch_32Msps", m.config4), ("config5_share_32_of_256ch_50Msps", m.config5_share)):
        name, codes, rows, n_stream, correlators, samples = build()
        ms = m.measure(codes, rows, n_stream, splits=0, device=dev_index)
        out[key] = {"workload": name, "ms_per_launch": ms, "correlators_per_s": correlators / (ms * 1e-3),
                    "channel_samples_per_s": samples / (ms * 1e-3)}
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import gnss_sdr_amd
    from gnss_sdr_amd.tracking import CorrelatorBank
    from gnss_sdr_amd.codes import gps_l1_ca_code
    import oracle  # the checker (spot check below) and the CPU baseline only; nothing the GPU legs consume comes from it

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if os.environ.get("GSH_BENCH_SHARE_GPU") == "1":
        local = 0  # self-test of the N > 1 code path on a one-GPU box (with GSH_BENCH_BACKEND=gloo): every rank uses GPU 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GSH_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for the one-GPU self-test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fs, n, C, E, T = a.fs, int(round(a.fs * 1e-3)), a.channels, a.epochs, a.taps
    n_samples = (E + 2) * n
    n_samples += (-n_samples) % 2

    # ---- inputs, resident in HBM before the timed region
    if rank == 0:
        x, dop, cph = make_stream_torch(torch, dev, n_samples, fs)
    else:
        x = torch.zeros(n_samples, dtype=torch.complex64, device=dev)
        dop, cph = np.zeros(0), np.zeros(0)
    cs = torch.cuda.Stream(device=dev)      # a real (non-null) stream: the engine's launches and RCCL's waits are ordered on it
    stream = cs.cuda_stream
    D = None
    # GSH_BENCH_FORCE_DIST=1 runs the N > 1 step structure (raw block -> convert -> correlate) on one GPU: a self-test of that path
    if world > 1 or os.environ.get("GSH_BENCH_FORCE_DIST") == "1":
        # N > 1: the shared IF stream reaches the other GPUs the way a front-end delivers it -- 8-bit I/Q (item_type ibyte,
        # 2 bytes per sample) -- and every rank converts it to complex64 on its own GPU (data_type_adapter arithmetic,
        # gsh_convert_samples_device) before correlating.  Distribution: scatter + all-gather over all xGMI links
        # (gnss_sdr_amd.sharding.BlockDistributor), double-buffered: block k+1 travels while block k is correlated.
        from gnss_sdr_amd.sharding import BlockDistributor
        from gnss_sdr_amd.sample_stream import convert_samples_device
        D = BlockDistributor(2 * n_samples, world, rank, 0, os.environ.get("GSH_BENCH_DIST", "scatter_allgather"))
        raw = [torch.zeros(D.padded, dtype=torch.int8, device=dev) for _ in range(2)]
        piece = [torch.zeros(D.chunk, dtype=torch.int8, device=dev) for _ in range(2)]
        raw_src = None
        if rank == 0:
            raw_src = torch.zeros(D.padded, dtype=torch.int8, device=dev)
            raw_src[:2 * n_samples] = torch.view_as_real(x).mul(30.0).round_().clamp_(-127, 127).to(torch.int8).reshape(-1)
        torch.cuda.synchronize()
        try:
            D.finish(D.start(raw[0], raw_src, piece[0]))
            torch.cuda.synchronize()
        except Exception as e:  # keep the run alive on a communicator that refuses scatter / all_gather_into_tensor
            if rank == 0:
                print(f"bench: scatter+all_gather distribution failed ({e}); falling back to broadcast", file=sys.stderr)
            D = BlockDistributor(2 * n_samples, world, rank, 0, "broadcast")
            D.finish(D.start(raw[0], raw_src, piece[0]))
            torch.cuda.synchronize()
    bank = CorrelatorBank(C, 1023, device=local)
    for c in range(C):
        bank.set_code(c, gps_l1_ca_code((rank * C + c) % 32 + 1))
    jobs, rows = build_jobs(C, E, n, fs, T, dop, cph, rank)
    bank.upload_jobs(jobs)
    bank.set_splits(1)
    bank.set_stream_device(x.data_ptr(), n_samples, keepalive=x)

    def step(k):
        if D is None:
            bank.launch(stream)
            return
        cur, nxt = k % 2, (k + 1) % 2
        works = D.start(raw[nxt], raw_src, piece[nxt])      # block k+1 on the communicator's stream
        convert_samples_device(local, raw[cur].data_ptr(), "ibyte", x.data_ptr(), n_samples, hip_stream=stream)
        bank.launch(stream)                                  # block k: convert, then correlate, on the compute stream
        D.finish(works)                                      # the compute stream waits for block k+1 before the next step reads it

    with torch.cuda.stream(cs):
        # set-up, not warm-up: bring the clocks out of their idle state with the same launches (a fixed count, identical on every rank,
        # and even, so that the double-buffer parity of step() is preserved)
        for k in range(2 * (max(a.settle_steps, 0) // 2)):
            step(k)
        for k in range(a.warmup):
            step(k)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(a.warmup + k)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: HIP events on the launch stream, inputs resident; taken straight after the timed region,
    # before the host-side spot check lets the GPU fall idle again
    k_ms = bank.time_launches(100)

    # ---- spot-check against the oracle (not timed): a few jobs of the last launch
    out = bank.read_outputs()
    if rank == 0:
        from helpers import oracle_job, scale_err
        xh = x.cpu().numpy()
        for j in (0, 1, C + 3, len(rows) - 1):
            o32, t64, sabs = oracle_job(oracle.ca_code(rows[j]["code_slot"] % 32 + 1), xh, rows[j])
            err = scale_err(out[j, :T], t64, sabs)
            if not np.all(err <= 1e-6):
                raise SystemExit(f"bench: GPU result of job {j} disagrees with the oracle: {out[j, :T]} vs {t64}")

    n_jobs = C * E
    alg_bytes = n_jobs * (8.0 * n + 8.0 * T)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("jobs") == n_jobs and tj.get("n") == n:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        total_corr = float(C) * T * E * a.steps * world
        res = {
            "metric": "correlators/s",
            "value": total_corr / dt,
            "unit": "correlators/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GPS L1 C/A tracking, {C} channels/GPU x {E} epochs/step, fs={fs / 1e6:g} Msps, N={n}, {T}-tap E/P/L, open-loop",
                       "channels_per_gpu": C, "epochs_per_step": E, "samples_per_epoch": n, "taps": T,
                       "parallelism": f"channels sharded over {world} GPU(s)" + (f", 8-bit stream block re-distributed over RCCL each step ({D.mode}, overlapped) and converted on every GPU" if D is not None else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": f"mcorr_kernel<{3 if T <= 3 else (5 if T <= 5 else 8)},0,false>",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_only_value": float(C) * T * E / (k_ms * 1e-3),
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(C, n, fs, T, a.cpu_seconds)
        if world == 1 and not a.no_acq:
            try:
                res["pcie_inclusive"] = pcie_inclusive_metric(torch, local, x, jobs, C, E, T, n_samples)
            except Exception as e:
                res["pcie_inclusive"] = {"error": str(e)}
            try:
                res["acquisition"] = acquisition_metric(torch, local, x[:n].contiguous(), fs)
            except Exception as e:
                res["acquisition"] = {"error": str(e)}
            try:
                res["closed_loop"] = closed_loop_metric(local, x, n_samples, fs, n, dop, cph, channels=C, epochs=min(E - 2, 200))
                # one compute unit per channel: 32 channels use an eighth of the chip, 256 (BASELINE config 5's channel count) fill it
                res["closed_loop_256ch"] = closed_loop_metric(local, x, n_samples, fs, n, dop, cph, channels=256, epochs=min(E - 2, 200))
            except Exception as e:
                res["closed_loop"] = {"error": str(e)}
            if not a.no_other_configs:
                try:
                    res["other_configs"] = other_configs_metric(local)
                except Exception as e:
                    res["other_configs"] = {"error": str(e)}
        print(json.dumps(res))
    bank.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
