#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: tracking correlators/s.

Workload (config.workload): BASELINE config 2 -- GPS L1 C/A, 32 channels (PRN 1..32), fs = 25 Msps,
N = 25 000 samples per 1 ms epoch, 3-tap E/P/L -- open-loop (pre-computed NCO parameter table), every channel
reading its own window sequence of ONE shared complex64 IF stream (noise + 8 embedded signals at 45 dB-Hz).
A "step" is one pass of the hot path over one batch of synthetic input: `--blocks-per-step` (1 400) consecutive blocks of the IF stream, each
block = channels x epochs jobs in one launch (the resident job table is re-used block after block through gsh_bank_set_sample_base).
Inputs (stream, codes, job table) are resident in HBM before the timed region; the stream cycles through a buffer of `--ring-blocks`
(8) blocks = 643 MB, larger than the 256 MiB Infinity Cache, so what the kernel does not find in L2 really comes from HBM.
value = channels*taps*epochs*blocks / time, whole job.  One step is ~0.25 s of GPU time: the driver's K = 20 steps keep the GPU busy for ~5 s,
long enough for its SMI samples to see it.  At N = 1 the blocks alternate over `--launches-in-flight` (2) correlator banks, each with a HIP stream of
its own: consecutive blocks are independent jobs, so the start of one launch fills the drain of the previous one and the gap between two launches of one
stream (172 -> 160 us per block, profiles/ab/r06/session30.txt); `value_single_stream` is one launch after the other on one stream, as `value` was until round 6,
and roofline.kernel_ms is one launch alone.

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU; every rank tracks its own 32 channels of the same
  stream (weak scaling).  Every block is replicated from the ingest GPU (rank 0) to the others in the front-end's 8-bit format by the
  engine itself -- gsh_stream_group_* (csrc/stream_group.hip: RCCL broadcast, or scatter + all-gather across all xGMI links with
  GSH_BENCH_DIST=scatter_allgather), converted to complex64 into every GPU's sample ring and correlated there; copies and kernels are
  ordered by events per sample range, so block k + 1 travels while block k is correlated.  All of that IS inside the timed region.
  torch.distributed only hands the 128-byte communicator id round and provides the barrier / MAX-reduce of the contract -- no data.

One JSON line on stdout (rank 0).  Besides the contract keys it carries
  roofline      -- dominant kernel (mcorr_kernel<3,0,false,false,false,true>: E/P/L, whole-code table, paired taps) against the roofline that BINDS it:
                   vector-ALU issue (bound "valu": SURVEY 8(d)'s algorithmic flops over the dense FP32 peak, frac <= 1); the contract's byte rate
                   (8N+8T per job over the HBM peak -- a rate above 1 because 32 channels share one stream) sits in roofline.contract_hbm;
                   duration from HIP events on the launch stream
  dropin        -- what a receiver gets through the reference's own seam: 32 dll_pll_veml_tracking_hip blocks, one thread each, ONE stream, ONE
                   Hip_Tracking_Runtime (tests/host/test_tracking_adapters bench): channel-periods/s through general_work, windows checked
                   against 32 reference blocks
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
FP32_PEAK_TFLOPS = 157.3  # dense packed-FP32 vector peak (256 CUs x 2.4 GHz x 256 flop/clk/CU), same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=400, help="1 ms epochs per channel per step")
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--taps", type=int, default=3)
    ap.add_argument("--blocks-per-step", type=int, default=1400, help="stream blocks (launches) one step works through (1 400 = ~0.25 s of GPU time)")
    ap.add_argument("--ring-blocks", type=int, default=8, help="blocks of stream resident in HBM that the steps cycle through (8 = 643 MB > Infinity Cache)")
    ap.add_argument("--settle-steps", type=int, default=3,
                    help="untimed steps run during set-up, before the W warm-up steps, so that the GPU clocks have settled: after an idle "
                         "period the first ~40 ms of work run up to 25 %% slower (profiles/ab/clock_ramp.py); 0 disables")
    ap.add_argument("--launches-in-flight", type=int, default=2,
                    help="N = 1: blocks alternate over this many correlator banks, each on a stream of its own (2: the start of one launch fills the drain of the "
                         "other and the gap between two launches of one stream, profiles/ab/r06/session30.txt); 1: one bank, one stream, as until round 6.  "
                         "value_single_stream is reported either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-acq", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in leg (32 tracking blocks through general_work)")
    ap.add_argument("--dropin-seconds", type=float, default=2.6, help="wall time of each drop-in run (a 2 400-period stream replayed seamlessly; the first 600 periods are set-up)")
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


def host_cpu_quota():
    """What the box actually grants this process of its logical CPUs: the cgroup CPU quota (cgroup v2 cpu.max, v1 cfs_quota / cfs_period) and the affinity mask --
    the reason `effective_parallelism` sits far below `host_logical_cpus` on a leased container."""
    out = {"affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None, "cgroup_cpus": None}
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["cgroup_cpu_max"] = f"{quota} {period}"
        out["cgroup_cpus"] = None if quota == "max" else float(quota) / float(period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_cpu_max"] = f"{q} {per}"
            out["cgroup_cpus"] = None if q <= 0 else q / per
        except Exception:
            pass
    return out


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

    # threads = what the box grants this process (cgroup quota / affinity), not its logical CPU count: 256 threads on a 16-CPU quota fight over the same cores and the
    # figure wanders (0.31 ... 0.64 M correlators/s across rounds).  The sweep 1, 2, 4, ..., quota (SURVEY 8d / BASELINE.md section 4) is reported beside it.
    q = host_cpu_quota()
    granted = min([v for v in (q.get("affinity_cpus"), q.get("cgroup_cpus"), os.cpu_count()) if v] or [1])
    quota_threads = max(1, min(channels, int(np.floor(granted + 1e-9)) or 1))
    sweep, cand = [], 1
    while True:
        threads = min(cand, quota_threads)
        e = max(16, 2 * threads * 16 // channels)
        run(e)                                    # warm-up at this thread count
        e = max(e, 64 if threads == 1 else e)
        rate = channels * e / max(run(e), 1e-9)
        sweep.append({"threads": threads, "value": rate * taps})
        if threads >= quota_threads:
            break
        cand *= 2
    rate_1 = sweep[0]["value"] / taps
    threads = quota_threads
    cores = threads
    epochs = 64
    t = run(epochs)  # calibration, then grow the sample until it fills about target_s of wall time
    while t < target_s / 3.0 and epochs < 200000:
        epochs = int(min(200000, max(epochs * 2, epochs * 0.9 * target_s / max(t, 1e-6))))
        t = run(epochs)
    simd = bool(R is not None and R.ref_simd_supported())
    rate_best = channels * epochs / t
    return {
        "value": channels * taps * epochs / t,
        "unit": "correlators/s",
        "cores": cores,                       # threads used
        "host_logical_cpus": os.cpu_count(),
        "host_cpu_quota": host_cpu_quota(),   # cgroup quota / affinity: what of those CPUs the container may use
        "single_thread_value": rate_1 * taps,
        "thread_sweep": sweep,                # correlators/s at 1, 2, 4, ..., quota threads (short samples)
        "best_of_sweep": max(sweep, key=lambda r: r["value"]),
        "effective_parallelism": (rate_best / rate_1) if rate_1 > 0 else None,   # what the threads delivered, in single-thread units
        "kind": kind,
        "sample": f"{channels} channels x {epochs} epochs of {n} samples, {taps} taps, {cores} threads, "
                  + ("Cpu_Multicorrelator_Real_Codes over volk_gnsssdr " + ("u_avx" if simd else "generic") + " protokernels (oracle/_ref)"
                     if kind == "reference" else "oracle/gnss_oracle.c scalar port"),
        "seconds": t,
    }


def acquisition_metric(torch, dev_index, x_block, fs, pmc=None):
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
    # GSH_BENCH_ACQ_SINGLE_STREAM=1 (profiles/run_profiles_r03.sh): no second batch in flight, so that a kernel trace shows every kernel's own duration
    # (two cell launches that share the compute units each take longer on the trace's clock than either alone)
    two = os.environ.get("GSH_BENCH_ACQ_SINGLE_STREAM", "0") != "1"
    acq.time_dwells(x_block, 32, reps=1500, pipelined=two)        # ~0.2 s untimed: the clocks settle (profiles/ab/clock_ramp.py)
    ms_serial = acq.time_dwells(x_block, 32, reps=20)             # one batch after the other on one stream: latency
    ms = acq.time_dwells(x_block, 32, reps=200, pipelined=two)    # batches alternating on two streams: throughput
    nbytes = 16.0 * n * 41 * (32 + 1)
    t = ms * 1e-3
    # (D + P D) (5 N log2 N + 6 N) flops per batch
    flops = (41 + 32 * 41) * (5.0 * n * np.log2(n) + 6.0 * n)
    unique = 8.0 * n + 8.0 * n * 32 + 16.0 * 32      # the input block, the 32 code spectra, the result records: what must come from HBM once
    roof = {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "frac_single_stream": nbytes / (ms_serial * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_is": "pipelined (two batches in flight)",
            "algorithmic_bytes_per_batch": nbytes, "kernels": "oc_forward_kernel + oc_cell_kernel<Plan<25,25,40>,false,false>",
            "binding": "valu+lds (one transform per compute unit: register butterflies, LDS exchanges between barriers)",
            "valu": {"algorithmic_flops_per_batch": flops, "achieved_tflops": flops / t / 1e12, "peak_tflops": FP32_PEAK_TFLOPS,
                     "frac": flops / t / 1e12 / FP32_PEAK_TFLOPS},
            "hbm_unique": {"bytes_per_batch": unique, "achieved_GBs": unique / t / 1e9, "frac": unique / t / 1e9 / HBM_PEAK_GBS}}
    m = (pmc or {}).get("acquisition")
    if m and m.get("n") == n and m.get("n_prn") == 32 and m.get("n_bins") == 41:
        roof["traffic"] = m.get("hbm_bytes_per_batch")
        roof["traffic_source"] = "static:" + str((pmc or {}).get("file")) + " (the builder's profiling run of this command)"
        roof["traffic_is"] = "memory-fabric bytes (FETCH_SIZE / WRITE_SIZE count Infinity Cache hits as well as HBM accesses): an upper bound of the HBM traffic"
        roof["pmc"] = {k: v for k, v in m.items() if k not in ("n", "n_prn", "n_bins")}
    res = {"metric": "acquisition dwells/s", "value": 32.0 / (ms * 1e-3), "unit": "dwells/s", "ms_per_batch": ms,
           "value_is": "PIPELINED: batches alternate on two streams, two in flight (throughput); value_single_stream is one batch after the other on one stream",
           "value_single_stream": 32.0 / (ms_serial * 1e-3), "ms_per_batch_single_stream": ms_serial,
           "config": {"workload": "GPS L1 C/A PCPS, 32 PRN x 41 Doppler bins, N=25000, 1 dwell"},
           "roofline": roof}
    acq.close()
    try:
        # beyond one compute unit's plan: the 50 000-point batch of BASELINE config 5's rate (50 Msps x 1 ms) on the split plan 2 x (25, 25, 40)
        n2 = 2 * n
        x2 = torch.view_as_complex(torch.randn(n2, 2, device=x_block.device).contiguous())
        acq2 = PcpsAcquisitionBank(fs_in=int(2 * fs), fft_size=n2, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(2 * fs / 1.023e6)),
                                   samples_per_code=float(n2), max_prn=32, device=dev_index, keep_grid=False)
        for p in range(32):
            acq2.set_local_code(p, gps_l1_ca_code_sampled(p + 1, int(2 * fs)))
        acq2.time_dwells(x2, 32, reps=100, pipelined=two)
        ms2_serial = acq2.time_dwells(x2, 32, reps=20)
        ms2 = acq2.time_dwells(x2, 32, reps=100, pipelined=two)
        res["split_plan_50000"] = {"workload": "32 PRN x 41 Doppler bins, N=50000 (50 Msps x 1 ms), 1 dwell, plan 2 x (25,25,40)", "ms_per_batch": ms2,
                                   "ms_per_batch_single_stream": ms2_serial, "value": 32.0 / (ms2 * 1e-3), "unit": "dwells/s",
                                   "algorithmic_GBs": 16.0 * n2 * 41 * 33 / (ms2 * 1e-3) / 1e9}
        acq2.close()
    except Exception as e:
        res["split_plan_50000"] = {"error": str(e)}
    try:
        # BASELINE config 4's acquisition length: Galileo E1, 4 ms at 32 Msps = 128 000 points on the split plan 5 x (25, 32, 32), decimation in time
        n4, fs4 = 128000, 32e6
        x4 = torch.view_as_complex(torch.randn(n4, 2, device=x_block.device).contiguous())
        acq4 = PcpsAcquisitionBank(fs_in=int(fs4), fft_size=n4, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=int(np.ceil(fs4 / 1.023e6)),
                                   samples_per_code=float(n4), max_prn=32, device=dev_index, keep_grid=False)
        rng4 = np.random.default_rng(4)
        for p in range(32):
            acq4.set_local_code(p, (rng4.integers(0, 2, n4) * 2 - 1).astype(np.complex64))   # +-1 replicas: the cost does not depend on the code values
        acq4.time_dwells(x4, 32, reps=20, pipelined=two)
        ms4_serial = acq4.time_dwells(x4, 32, reps=10)
        ms4 = acq4.time_dwells(x4, 32, reps=20, pipelined=two)
        res["split_plan_128000"] = {"workload": "32 PRN x 41 Doppler bins, N=128000 (Galileo E1 4 ms at 32 Msps), 1 dwell, plan 5 x (25,32,32), decimation in time",
                                    "ms_per_batch": ms4, "ms_per_batch_single_stream": ms4_serial, "value": 32.0 / (ms4 * 1e-3), "unit": "dwells/s",
                                    "algorithmic_GBs": 16.0 * n4 * 41 * 33 / (ms4 * 1e-3) / 1e9}
        acq4.close()
    except Exception as e:
        res["split_plan_128000"] = {"error": str(e)}
    # CPU baselines: the reference's own block (kind "reference"; its transform here is oracle/ref_fft.cc, not FFTW), and the numpy / pocketfft restatement
    # (kind "port": the faster transform, not the reference's code) -- both reported, neither is the target
    try:
        res["cpu_baseline"] = acquisition_reference_baseline(x_block.cpu().numpy(), fs, n)
    except Exception as e:
        res["cpu_baseline"] = {"error": str(e)}
    try:
        res["cpu_baseline_port"] = acquisition_cpu_baseline(x_block.cpu().numpy(), fs, n)
    except Exception as e:
        res["cpu_baseline_port"] = {"error": str(e)}
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


def acquisition_reference_baseline(x, fs, n, target_s=8.0):
    """SURVEY.md 8d (ii), kind "reference": the reference's OWN pcps_acquisition block (oracle/_ref/libgnsssdr_ref_acq.so: pcps_acquisition.cc compiled in place,
    acq.cc:522-560 through general_work, :749-853) with the CPU FFT we ship behind gr::fft (oracle/ref_fft.cc: mixed radix, double precision -- FFTW and VOLK are not
    available in this image, so the transform is slower than the reference's own; the block's loops, wipe-off, magnitude and statistics are the reference's).  One
    thread per PRN like the reference's channels; one dwell = 41 Doppler bins over one 1 ms block."""
    import threading
    from oracle import ref_acq
    if not ref_acq.available():
        return {"error": "oracle/_ref/libgnsssdr_ref_acq.so was not built (needs /root/reference at build time)"}
    import oracle
    role = "Acquisition_1C"
    props = {"GNSS-SDR.internal_fs_sps": int(fs), role + ".blocking": "true", role + ".doppler_max": 5000, role + ".doppler_step": 250, role + ".max_dwells": 1,
             role + ".pfa": 0.001}
    x1 = np.ascontiguousarray(x[:n], np.complex64)

    def make(prn):
        b = ref_acq.RefAcqBlock(ref_acq.K_PCPS, props, 1.023e6, 2e6, 1, role=role, prn=prn, signal="1C", system="G")
        b.set_local_code(oracle.ca_code_complex_sampled(prn, int(fs)))
        return b

    def dwell(b):
        b.set_active(True)
        pos = 0
        for _ in range(8):  # state 0 -> 1 -> the dwell: a few general_work calls, as the scheduler makes them
            _, c = b.work(x1[pos:] if pos < n else x1)
            pos = (pos + c) % n
            if not b.status()["active"]:
                return

    b0 = make(1)
    dwell(b0)                       # warm-up: FFT plan, page faults
    t0 = time.perf_counter()
    dwell(b0)
    t1 = time.perf_counter() - t0   # one dwell on one thread
    b0.close()
    threads = min(os.cpu_count() or 1, 32)   # one per PRN, as the reference's channels
    reps = int(max(1, min(200, target_s / max(t1, 1e-3))))
    blocks = [make(p + 1) for p in range(threads)]
    for b in blocks:
        dwell(b)
    done = [0] * threads

    def work(i):
        for _ in range(reps):
            dwell(blocks[i])
            done[i] += 1

    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for b in blocks:
        b.close()
    n_dwells = sum(done)
    return {"value": n_dwells / dt, "unit": "dwells/s", "cores": threads, "kind": "reference",
            "sample": f"{n_dwells} dwells (1 PRN x 41 bins x {n} samples each) by {threads} threads, each driving its own pcps_acquisition block (the reference's source, "
                      f"compiled in place) through general_work; gr::fft = oracle/ref_fft.cc (no FFTW in this image)",
            "single_thread_dwells_per_s": 1.0 / t1, "seconds": dt}


def pcie_inclusive_metric(torch, dev_index, x_dev, jobs, C, E, T, n_samples, reps=24):
    """Not `value`: the same block when the stream arrives from HOST memory -- 8-bit items (what a front-end delivers) in a pinned buffer.
    sequential: gsh_stream_push (H2D + cast, host waits) -> launch -> wait, block after block (round 1's figure);
    overlapped: gsh_stream_push_async keeps block k + 1 on the PCIe bus while block k is correlated (events per sample range order the
                ring's writer and the correlator; the host only waits at the very end)."""
    from gnss_sdr_amd.sample_stream import SampleStream
    from gnss_sdr_amd.tracking import CorrelatorBank
    from gnss_sdr_amd.codes import gps_l1_ca_code
    q = torch.view_as_real(x_dev[:n_samples]).mul(30.0).round_().clamp_(-127, 127).to(torch.int8).cpu()
    host = torch.empty_like(q).pin_memory()
    host.copy_(q)
    h = host.numpy()
    out = {}
    for mode in ("sequential", "overlapped"):
        ring = SampleStream(3 * n_samples + 2, n_samples // 2, device=dev_index)
        bank = CorrelatorBank(C, 1023, device=dev_index)
        for c in range(C):
            bank.set_code(c, gps_l1_ca_code(c % 32 + 1))
        bank.set_stream_ring(ring)
        bank.set_splits(1)
        assert ring.push(h, "ibyte") == 0       # block 0 resident: the job table (window positions relative to a block start) is staged once
        bank.upload_jobs(jobs)
        bank.launch()
        bank.synchronize()
        t0 = time.perf_counter()
        if mode == "sequential":
            for r in range(1, reps + 1):
                first = ring.push(h, "ibyte")
                bank.set_sample_base(first)
                bank.launch()
                bank.synchronize()
        else:
            first = ring.push_async(h, "ibyte")
            for r in range(1, reps + 1):
                nxt = ring.push_async(h, "ibyte") if r < reps else None   # block r + 1 on its way ...
                bank.set_sample_base(first)
                bank.launch()                                               # ... while block r is correlated
                first = nxt
            bank.synchronize()
            ring.wait()
        dt = (time.perf_counter() - t0) / reps
        out[mode] = {"value": C * T * E / dt, "ms_per_block": dt * 1e3}
        bank.close()
        ring.close()
    return {"value": out["overlapped"]["value"], "unit": "correlators/s", "ms_per_step": out["overlapped"]["ms_per_block"],
            "sequential": out["sequential"], "host_bytes_per_block": int(h.nbytes),
            "note": "8-bit stream block from pinned host memory -> device ring (H2D + cast) -> launch over the resident job table; `value`: "
                    "gsh_stream_push_async, next block's copy overlapped with this block's correlation; `sequential`: nothing overlapped"}


def closed_loop_metric(dev_index, x_dev, n_samples, fs, n, dop, cph, channels=32, epochs=200, lock_detectors=False, live=False, split=1):
    """Secondary metric: the DLL/PLL loop closed on the device (gsh_trk_*), BASELINE config 2 shape -- every channel runs
    `epochs` consecutive code periods with its own discriminators / loop filters / NCO update between them, one launch."""
    try:
        from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    except Exception as e:
        return {"error": f"tracking loop unavailable: {e}"}
    from gnss_sdr_amd.codes import gps_l1_ca_code
    # lock_detectors: cn0_and_tracking_lock_status on the device as well (what the tracking adapters run with); the fail limits are out of reach so that
    # the channels without a signal keep running for the timing
    conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0, enable_lock_detectors=int(lock_detectors), max_carrier_lock_fail=1 << 30,
                    max_code_lock_fail=1 << 30)
    loop = TrackingLoop(conf, channels, 1023, device=dev_index)
    loop.set_stream_device(x_dev.data_ptr(), n_samples, keepalive=x_dev)
    if split > 1:
        loop.set_split(split)  # gsh_trk_set_split: `split` cooperating work-groups share every window of a channel
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
    out = {"metric": "correlators/s, loop closed on device", "value": channels * 3 * epochs / (ms * 1e-3), "unit": "correlators/s",
           "ms_per_launch": ms, "us_per_epoch": ms * 1e3 / epochs, "channels": channels, "epochs_per_launch": epochs, "lock_detectors": bool(lock_detectors),
           "work_groups_per_channel": split,
           "channels_with_signal_locked": f"{locked}/{min(channels, len(dop))}",
           "real_time_factor": epochs * 1e-3 / (ms * 1e-3)}
    # what binds it: the correlation's vector instructions (6 + 4 T flops per channel-sample) on the compute units the channels occupy -- one each
    flops = float(channels) * epochs * (6 + 4 * 3) * n
    peak = 157.3e12
    out["roofline"] = {"bound": "valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / peak,
                       "frac_of_the_compute_units_in_use": flops / (ms * 1e-3) / (peak * min(channels, 256) / 256.0),
                       "note": "one work-group (one compute unit) per channel; a period is a dependent chain: window -> 16-wave sums -> loop arithmetic on lane 0 of four waves -> "
                               "thread 0's join / update_tracking_vars -> next window; counters (profiles/ab/r04/closed_loop_steps.txt): the SIMDs issue vector instructions "
                               "72 % of the period, ~90 % of the correlation, a third of which is per-wave fixed cost (phasor set-up, wave sums)"}
    if live:
        # (the residency's 150 timed periods last 1.2 ms and are timed by a host loop that polls 32 record rings: one scheduling hiccup of that loop is 50 % -- the best of three)
        tries = [closed_loop_live(dev_index, x_dev, n_samples, fs, n, dop, cph, channels, epochs, conf) for _ in range(3)]
        good = [t for t in tries if "us_per_epoch" in t]
        out["live"] = min(good, key=lambda t: t["us_per_epoch"]) if good else tries[0]
        if good:
            out["live"]["us_per_epoch_of_the_three_residencies"] = [round(t["us_per_epoch"], 3) for t in good]
    return out


def closed_loop_config4_metric(torch, dev_index, channels=50, epochs=60, split=1):
    """BASELINE config 4 with the loop closed on the device: Galileo E1, 50 channels, fs 32 Msps, 4 ms windows of 128 000 samples, VE/E/P/L/VL on the pilot (E1C) + the data
    prompt (E1B) = 5 + 1 correlators per channel-period (trk.cc:1246-1256), lock detectors on.  One launch of `epochs` periods per channel over a resident stream
    (noise + four E1 signals at 45 dB-Hz so that some channels really track; the others run noise-driven, which costs the same)."""
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import add_code_signal, cn0_to_amplitude, golden_e1_l5_codes
    fs, n = 32e6, 128000
    g = golden_e1_l5_codes()
    n_stream = (epochs + 3) * n
    gen = torch.Generator(device=f"cuda:{dev_index}").manual_seed(0x5EED0004)
    x = torch.view_as_complex(torch.randn(n_stream, 2, device=f"cuda:{dev_index}", generator=gen).contiguous())
    sig = [(0, -1830.0, 3000.0), (11, 2410.0, 511.0), (23, 655.0, 7000.5), (37, -3120.0, 123.0)]
    add = np.zeros(n_stream, np.complex64)
    amp = cn0_to_amplitude(45.0, fs)
    starts = {}
    for ch, fd, ph in sig:
        rate = 1.023e6 * (1 + fd / 1575.42e6) / fs * 2.0
        add_code_signal(add, (g["e1b"][ch] - g["e1c"][ch]) / np.sqrt(2.0), fs, rate, ph, fd, amp)
        starts[ch] = (int(round((8184.0 - ph) / rate)), fd)
    x += torch.from_numpy(add).to(x.device)
    conf = trk_conf(fs_in=fs, vector_length=n, code_length_chips=4092, code_samples_per_chip=2, veml=1, track_pilot=1, cloop=0, early_late_space_chips=0.15,
                    very_early_late_space_chips=0.5, pll_bw_hz=15.0, dll_bw_hz=0.75, pll_filter_order=3, dll_filter_order=2, enable_lock_detectors=1,
                    max_carrier_lock_fail=1 << 30, max_code_lock_fail=1 << 30)
    loop = TrackingLoop(conf, channels, 8184, device=dev_index)
    loop.set_stream_device(x.data_ptr(), n_stream, keepalive=x)
    if split > 1:
        loop.set_split(split)
    rng = np.random.default_rng(0x5EED0007)
    for c in range(channels):
        if c in starts:
            loop.start(c, g["e1c"][c], starts[c][0], 0, starts[c][1] - 5.0, data_code=g["e1b"][c])
        else:
            loop.start(c, g["e1c"][c % 50], int(rng.integers(0, n)), 0, float(rng.uniform(-4000, 4000)), data_code=g["e1b"][c % 50])
    loop.time_run(epochs, reps=3)
    ms = loop.time_run(epochs, reps=5)
    rec, done = loop.run(epochs)
    locked = sum(1 for c in starts if c < channels and abs(np.mean([r.carrier_doppler_hz for r in rec[c][-10:]]) - starts[c][1]) < 3.0)
    loop.close()
    corr = 6  # five pilot taps + the data prompt
    flops = float(channels) * epochs * ((6 + 4 * 5) + (6 + 4 * 1)) * n   # SURVEY 8d's figure per correlator call: 6 + 4 T flops per sample, pilot call + data call
    peak = 157.3e12
    return {"metric": "correlators/s, loop closed on device, BASELINE config 4", "workload": f"Galileo E1, {channels} channels, fs 32 Msps, 128000-sample (4 ms) windows, 5 + 1 taps, lock detectors on",
            "value": channels * corr * epochs / (ms * 1e-3), "unit": "correlators/s", "ms_per_launch": ms, "us_per_epoch": ms * 1e3 / epochs, "channels": channels,
            "epochs_per_launch": epochs, "work_groups_per_channel": split, "channels_with_signal_locked": f"{locked}/{sum(1 for c in starts if c < channels)}",
            "real_time_factor": epochs * 4e-3 / (ms * 1e-3),
            "roofline": {"bound": "valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / peak,
                         "frac_of_the_compute_units_in_use": flops / (ms * 1e-3) / (peak * min(channels * split, 256) / 256.0),
                         "note": "one compute unit per channel and cooperating work-group (50 x split of 256); a 4 ms period is 128000 samples x (5 pilot taps + the data prompt) = 62.5 trips of 2048 samples per wave"}}


def closed_loop_live(dev_index, x_dev, n_samples, fs, n, dop, cph, channels, epochs, conf):
    """The same channels in LIVE mode (gsh_trk_live_*): one residency of the loop kernel follows a ring that already holds the block; the records are read
    from page-locked host memory while it runs.  Rate = the slowest channel's progress between two marks."""
    from gnss_sdr_amd.codes import gps_l1_ca_code
    from gnss_sdr_amd.sample_stream import SampleStream
    from gnss_sdr_amd.tracking_loop import TrackingLoop
    ring = SampleStream(n_samples + 2 * n, 2 * n, device=dev_index)
    ring.push_device(x_dev.data_ptr(), n_samples)
    loop = TrackingLoop(conf, channels, 1023, device=dev_index)
    loop.set_stream_ring(ring)
    rng = np.random.default_rng(0x5EED0006)
    for c in range(channels):
        if c < len(dop):
            f_code = 1.023e6 * (1 + dop[c] / 1575.42e6)
            loop.start(c, gps_l1_ca_code(c + 1), int(round((1023.0 - cph[c]) / f_code * fs)), 0, float(dop[c]) + rng.uniform(-20, 20))
        else:
            loop.start(c, gps_l1_ca_code(c % 32 + 1), int(rng.integers(0, n)), 0, float(rng.uniform(-5000, 5000)))
    loop.live_configure(idle_timeout_us=2000, residency_us=2000000)
    lo, hi = 30, epochs - 20
    t_lo = t_hi = None
    loop.live_begin()
    watch = list(range(0, channels, max(1, channels // 32)))  # (a look costs Python a few microseconds per channel)
    t_end = time.perf_counter() + 5.0
    while t_hi is None and time.perf_counter() < t_end:
        p = min(loop.live_take(c, 0)[1] for c in watch)
        now = time.perf_counter()
        if t_lo is None and p >= lo:
            t_lo, p_lo = now, p
        if p >= hi:
            t_hi, p_hi = now, p
    loop.live_quiesce()
    loop.close()
    ring.close()
    if t_hi is None or t_lo is None or p_hi <= p_lo:
        return {"error": "the residency did not finish its periods within 5 s"}
    us = (t_hi - t_lo) * 1e6 / (p_hi - p_lo)
    return {"us_per_epoch": us, "value": channels * 3 / (us * 1e-6), "unit": "correlators/s", "periods_timed": p_hi - p_lo,
            "note": "one residency, no launch per batch of periods; records written to a ring in page-locked host memory as they finish"}


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
    try:
        out["mcorr16_config2_shape"] = mcorr16_metric(dev_index)
    except Exception as e:  # the 16-bit family is a side figure: never the reason the line is missing
        out["mcorr16_config2_shape"] = {"error": str(e)}
    return out


def mcorr16_metric(dev_index, channels=32, epochs=400, n=25000):
    """The 16-bit correlator family (Cpu_Multicorrelator_16sc's arithmetic, gsh_bank16_*) at BASELINE config 2's shape: complex int16 stream and codes, E/P/L,
    12 800 jobs per launch; four jobs of the launch checked bit for bit against the oracle, the reference's own object timed on one core when oracle/_ref travelled."""
    import numpy as np
    import torch
    from gnss_sdr_amd.tracking16 import CorrelatorBank16, make_job16
    import oracle
    rng = np.random.default_rng(0x16)
    x = rng.integers(-50, 51, size=((epochs + 1) * n, 2)).astype(np.int16)
    xd = torch.from_numpy(x).to(torch.device("cuda", dev_index))
    codes = [np.stack([oracle.ca_code(c + 1), np.zeros(1023, np.float32)], -1).astype(np.int16) for c in range(channels)]
    bank = CorrelatorBank16(channels, 1023, device=dev_index)
    for c in range(channels):
        bank.set_code(c, codes[c])
    bank.set_stream_device(xd.data_ptr(), len(x), keepalive=xd)
    shifts = np.array([-0.5, 0.0, 0.5], np.float32)
    jobs, plain = [], []
    for e in range(epochs):
        for c in range(channels):
            par = (float(np.float32(rng.uniform(0, 6.28))), float(np.float32(2 * np.pi * rng.uniform(-5000, 5000) / 25e6)), float(np.float32(rng.uniform(0, 1))), float(np.float32(1.023e6 / 25e6)))
            off = e * n + int(rng.integers(0, n))
            jobs.append(make_job16(off, n, c, *par, shifts))
            plain.append((off, c, par))
    bank.upload(jobs)
    ms = min(bank.time_launches(5) for _ in range(2))
    got = bank.read()
    ok = True
    for j in (0, len(jobs) // 3, len(jobs) // 2 + 7, len(jobs) - 1):
        off, c, par = plain[j]
        ok = ok and bool(np.array_equal(got[j, :3], oracle.mcorr16(codes[c], shifts, x[off:off + n], *par)))
    bank.close()
    res = {"workload": "32 ch x 400 epochs x 25000 samples, complex int16, E/P/L (Cpu_Multicorrelator_16sc's arithmetic)", "ms_per_launch": ms,
           "correlators_per_s": len(jobs) * 3 / (ms * 1e-3), "spot_check_bit_exact_vs_oracle": ok, "dtype": "s16 (float32 rotation)"}
    R = oracle.ref()
    if R is not None and hasattr(R, "ref_mcorr16_time"):
        out = np.zeros((3, 2), np.int16)
        for name, simd in (("cpu_reference_generic_us_per_call", 0), ("cpu_reference_simd_us_per_call", 1)):
            R.ref_set_flavour(simd)
            s = R.ref_mcorr16_time(codes[0].reshape(-1), 1023, shifts, 3, x.reshape(-1), len(x), n, 100, 0.3, 0.001, 0.2, float(np.float32(1.023e6 / 25e6)), out.reshape(-1))
            res[name] = s / 100 * 1e6
        R.ref_set_flavour(0)
    return res


def dropin_metric(channels, fs, periods, periods_per_call=20, seconds=0.0):
    """What a receiver gets through the reference's own seam (north_star: "drops into a Channel unchanged"): `channels` dll_pll_veml_tracking_hip
    blocks behind their TrackingInterface adapters, one scheduler thread each as in a flowgraph (gnss_flowgraph.cc:1227-1231), ONE 25 Msps stream,
    ONE Hip_Tracking_Runtime whose launches advance every channel that has samples.  The C++ program (tests/host/test_tracking_adapters bench,
    compiled against the reference's headers where /root/reference is present; the binary travels with the tree) drives general_work, times the
    threads, and checks every block's window positions against the reference's own block over the same stream."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "test_tracking_adapters")
    if not os.path.exists(exe):
        return {"error": "tests/host/test_tracking_adapters was not prebuilt (needs /root/reference at build time)"}
    # seconds > 0: a 2 400-period stream replayed seamlessly for that long (the steady window then lasts seconds, not tens of milliseconds); periods = the cap
    r = subprocess.run([exe, "bench", str(channels), str(int(fs)), str(periods), str(periods_per_call)] + ([f"{seconds:g}"] if seconds > 0 else []), capture_output=True,
                       text=True, timeout=900, cwd="/tmp")
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_JSON")]
    if r.returncode != 0 or not line:
        return {"error": (r.stdout[-600:] + r.stderr[-400:]).strip()}
    d = json.loads(line[-1][len("DROPIN_JSON"):])
    d["unit"] = "channel-periods/s through general_work (x3 taps = correlators/s)"
    return d


def sharded_legs(torch, cp, dev, local, rank, world, ring, block, fs, n, C, epochs=200):
    """The two other paths a receiver runs, on the rings of the stream group (every rank calls this; rank 0 gets the figures): SURVEY.md 8e --
    closed loop: channel c -> GPU c mod G, here C channels per GPU (weak scaling), each GPU's loop bound to ITS ring of the replicated stream;
    acquisition: PRN p -> GPU p mod G over the same replicated 1 ms block (strong scaling: 32 PRNs x 41 bins in all, 32 / G per GPU).
    Times are HIP-event / wall times per rank, reduced with MAX over the ranks like the headline."""
    from gnss_sdr_amd.acquisition import PcpsAcquisitionBank
    from gnss_sdr_amd.codes import gps_l1_ca_code, gps_l1_ca_code_sampled
    from gnss_sdr_amd.tracking_loop import TrackingLoop, trk_conf
    out = {}

    from gnss_sdr_amd.sharding import prns_of, weak_channel_prn
    reduce_max = cp.reduce_max

    lo, hi = ring.range()
    base = hi - block  # the newest whole block in the ring
    try:
        conf = trk_conf(fs_in=fs, vector_length=n, pll_bw_hz=35.0, dll_bw_hz=2.0)
        loop = TrackingLoop(conf, C, 1023, device=local)
        loop.set_stream_ring(ring)
        rng = np.random.default_rng(0x5EED0006 + rank)
        for c in range(C):
            loop.start(c, gps_l1_ca_code(weak_channel_prn(rank, C, c)), base + int(rng.integers(0, n)), base, float(rng.uniform(-5000, 5000)))
        loop.time_run(epochs, reps=10)
        ms = reduce_max(loop.time_run(epochs, reps=5))
        loop.close()
        out["closed_loop_sharded"] = {"metric": "correlators/s, loop closed on device, channels sharded over the GPUs", "value": float(world) * C * 3 * epochs / (ms * 1e-3),
                                      "unit": "correlators/s", "ms_per_launch": ms, "us_per_epoch": ms * 1e3 / epochs, "channels_per_gpu": C, "n_gpus": world,
                                      "scaling": "weak"}
    except Exception as e:
        out["closed_loop_sharded"] = {"error": str(e)}
        reduce_max(0.0)
    try:
        mine = prns_of(rank, world, 32)
        acq = PcpsAcquisitionBank(fs_in=int(fs), fft_size=n, doppler_max=5000, doppler_step=250, samples_per_chip=int(np.ceil(fs / 1.023e6)), samples_per_code=float(n),
                                  max_prn=max(len(mine), 1), num_doppler_bins=41, device=local)
        for k, p in enumerate(mine):
            acq.set_local_code(k, gps_l1_ca_code_sampled(p, int(fs)))
        for _ in range(5):
            acq.dwell_ring(ring, base + 7, max(len(mine), 1))
        reps = 50
        torch.cuda.synchronize()
        cp.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            acq.dwell_ring(ring, base + 7, max(len(mine), 1))
        dt = reduce_max(time.perf_counter() - t0)
        acq.close()
        out["acquisition_sharded"] = {"metric": "acquisition dwells/s (1 PRN x 41 Doppler bins), PRNs sharded over the GPUs", "value": 32.0 * reps / dt, "unit": "dwells/s",
                                      "ms_per_batch_of_32_prn": dt / reps * 1e3, "prn_per_gpu": len(mine), "n_gpus": world, "scaling": "strong",
                                      "note": "wall time of synchronous dwells out of the ring (results read back every call), not the pipelined kernel rate of the N = 1 leg"}
    except Exception as e:
        out["acquisition_sharded"] = {"error": str(e)}
        reduce_max(0.0)
    return out


def rccl_one_rank_metric(torch, dev_index, x_dev, block):
    """The stream group with ONE rank forced through RCCL (GSH_GROUP_FORCE_RCCL): ncclCommInitRank(1 rank), then every 8-bit block through ncclBroadcast, or the
    grouped ncclSend / ncclRecv to itself + ncclAllGather, on the ring's stream, cast into the ring behind it.  All the RCCL evidence a single-GPU box can give
    (the pool has no multi-GPU node): the ring must come out bit-identical to a group that exchanges nothing."""
    import numpy as np
    from gnss_sdr_amd.sample_stream import StreamGroup
    raw = torch.view_as_real(x_dev[:block]).mul(30.0).round_().clamp_(-127, 127).to(torch.int8).reshape(-1).contiguous()
    torch.cuda.synchronize()
    out = {"rccl_ranks": 1, "block_samples": int(block)}
    plain = StreamGroup.from_rank(dev_index, 0, 1, None, 3 * block + 2, block // 2, "broadcast")
    plain.push_device(raw.data_ptr(), block, "ibyte")
    plain.wait()
    want = plain.ring(0).read(block - block // 2, block // 2).view(np.uint32)
    for mode in ("broadcast", "scatter_allgather"):
        g = StreamGroup.from_rank(dev_index, 0, 1, None, 3 * block + 2, block // 2, mode, force_rccl=True)
        first = g.push_device(raw.data_ptr(), block, "ibyte")
        g.wait()
        same = bool(np.array_equal(g.ring(0).read(first + block - block // 2, block // 2).view(np.uint32), want))
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            g.push_device(raw.data_ptr(), block, "ibyte")
        g.wait()
        dt = (time.perf_counter() - t0) / reps
        info = g.rccl_info()
        out[mode] = {"ring_identical_to_a_group_without_exchange": same, "rccl_calls": info["collectives"], "ms_per_block": dt * 1e3,
                     "raw_GBs_through_the_collectives": 2.0 * block / dt / 1e9}
        out["rccl_version"] = info["version"]
        out["communicator_ranks"] = info["ranks"]
        g.close()
    plain.close()
    return out


def bench_summary(res):
    """The headline figures of every leg once more, short, at the END of the line (the driver keeps the tail of stdout)."""
    def get(d, *keys):
        for k in keys:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    s = {"tracking_Mcorr_s": res["value"] / 1e6, "tracking_launches_in_flight": res.get("launches_in_flight"),
         "tracking_Mcorr_s_single_stream": (res.get("value_single_stream") or 0) / 1e6, "tracking_kernel_ms": get(res, "roofline", "kernel_ms"), "tracking_valu_frac": get(res, "roofline", "frac"),
         "acq_dwells_s_pipelined": get(res, "acquisition", "value"), "acq_dwells_s_single_stream": get(res, "acquisition", "value_single_stream"),
         "acq_ms_per_batch_pipelined": get(res, "acquisition", "ms_per_batch"), "acq_ms_per_batch_single_stream": get(res, "acquisition", "ms_per_batch_single_stream"),
         "acq_hbm_frac_pipelined": get(res, "acquisition", "roofline", "frac"), "acq_hbm_frac_single_stream": get(res, "acquisition", "roofline", "frac_single_stream"),
         "acq_128000_ms": get(res, "acquisition", "split_plan_128000", "ms_per_batch"), "acq_50000_ms": get(res, "acquisition", "split_plan_50000", "ms_per_batch"),
         "hbm_read_probe_GBs": res.get("hbm_read_probe_GBs"), "hbm_unique_frac": res.get("hbm_unique_frac"), "valu_issue_frac": res.get("valu_issue_frac"),
         "cpu_baseline_threads": get(res, "cpu_baseline", "cores"),
         "closed_loop_us": get(res, "closed_loop", "us_per_epoch"), "closed_loop_detectors_us": get(res, "closed_loop_lock_detectors", "us_per_epoch"),
         "closed_loop_live_us": get(res, "closed_loop_lock_detectors", "live", "us_per_epoch"), "closed_loop_256ch_us": get(res, "closed_loop_256ch", "us_per_epoch"),
         "closed_loop_config4_us": get(res, "closed_loop_config4", "us_per_epoch"),
         "closed_loop_config2_2wg_us": get(res, "closed_loop_cooperating", "config2_2_work_groups", "us_per_epoch"),
         "closed_loop_config4_4wg_us": get(res, "closed_loop_cooperating", "config4_4_work_groups", "us_per_epoch"),
         "mcorr16_Mcorr_s": (get(res, "other_configs", "mcorr16_config2_shape", "correlators_per_s") or 0) / 1e6,
         "dropin_20_Mcps": get(res, "dropin", "value"), "dropin_1_Mcps": get(res, "dropin", "one_period_per_call", "value"),
         "rccl_ranks_exercised": get(res, "rccl_one_rank", "communicator_ranks"), "rccl_version": get(res, "rccl_one_rank", "rccl_version"),
         "cpu_baseline_Mcorr_s": (get(res, "cpu_baseline", "value") or 0) / 1e6}
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()}


def load_pmc():
    """profiles/pmc_r02.json: per-launch counter averages of the dominant kernels under this very command (profiles/run_profiles_r02.sh +
    profiles/summarize_r02.py; rocprofv3 --pmc passes, kernel-trace only).  None when absent."""
    for name in ("pmc_r06.json", "pmc_r05.json", "pmc_r04.json", "pmc_r03.json", "pmc_r02.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            d["file"] = "profiles/" + name
            return d
        except Exception:
            continue
    return None


def tracking_roofline(C, E, T, n, k_ms, pmc):
    """The roofline object for mcorr_kernel (DESIGN.md section 7 has the formulas).
    bound / achieved / peak / frac: what BINDS the kernel -- vector-ALU issue (the float32 chip-index arithmetic of the taps): SURVEY
      8(d)'s algorithmic flops (6 + 4T per channel-sample) per launch / kernel_ms over the dense FP32 vector peak; a utilisation (<= 1).
    traffic: HBM bytes per launch from the PMC counters of the committed profile (pmc_static: collected by profiles/run_profiles_r03.sh under
      this very command, not re-measured in this run).
    contract_hbm: SURVEY 8(d)'s ALGORITHMIC bytes (8N + 8T per channel-epoch: every channel charged a private read of its window) over the HBM
      peak.  32 channels share one stream, so this rate exceeds 1 -- it is not a utilisation.
    hbm_unique: the bytes that must leave HBM once per launch (the stream block, the codes, the results) over the HBM peak -- the honest HBM figure."""
    n_jobs = C * E
    alg_bytes = n_jobs * (8.0 * n + 8.0 * T)
    t = k_ms * 1e-3
    achieved = alg_bytes / t / 1e9
    flops = float(C) * E * n * (6.0 + 4.0 * T)
    unique = 8.0 * (E + 1) * n + 4.0 * 1023 * C + 64.0 * n_jobs
    tflops = flops / t / 1e12
    r = {"bound": "valu", "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS, "traffic": None,
         "kernel": (f"mcorr_kernel_t128<3,0,false,false,false,true> (two waves per job: launches of >= 5 120 E/P/L jobs, csrc/multicorrelator_t128.hip)"
                    if (T == 3 and n_jobs >= 5120) else f"mcorr_kernel<{3 if T <= 3 else (5 if T <= 5 else 8)},0,false,false,false,{'true' if T == 3 else 'false'}>"),
         "kernel_ms": k_ms, "traffic_source": None,
         "algorithmic_flops_per_launch": flops,
         "note": "bound by vector-ALU issue (the float32 chip-index chains of the taps), not by HBM: 32 channels read ONE stream, the block is fetched from HBM once "
                 "and served from L2 to the other 31; MFMA does not apply (per-channel mat-vec, f32 MFMA runs at the vector rate on gfx950)",
         "contract_hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "rate_over_peak": achieved / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": alg_bytes,
                          "note": "SURVEY 8(d)'s per-unit bytes (8N + 8T per channel-epoch: every channel charged a private read of its window); a rate, "
                                  "not a utilisation -- it exceeds 1 because the stream is shared"},
         "hbm_unique": {"bytes_per_launch": unique, "achieved_GBs": unique / t / 1e9, "frac": unique / t / 1e9 / HBM_PEAK_GBS},
         "channel_samples_per_s": float(C) * E * n / t}
    m = (pmc or {}).get("mcorr")
    if m and m.get("jobs") == n_jobs and m.get("n") == n:
        r["traffic"] = m.get("hbm_bytes_per_launch")
        r["traffic_source"] = "static:" + str((pmc or {}).get("file")) + " (rocprofv3 --pmc passes of the builder's profiling run of this command, not counters of THIS run)"
        if m.get("SQ_INSTS_VALU") and m.get("kernel_avg_us"):
            # share of the vector-ALU issue slots the launch used: 4 clocks per wave64 instruction over 1 024 SIMDs x the kernel's clocks (static counters, 2.4 GHz)
            r["valu_issue_frac"] = 4.0 * m["SQ_INSTS_VALU"] / (1024.0 * m["kernel_avg_us"] * 1e-6 * 2.4e9)
            r["valu_issue_frac_source"] = r["traffic_source"]
        r["pmc_static"] = {k: m[k] for k in ("source", "kernel_avg_us", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
                                      "valu_insts_per_channel_sample", "valu_cycles_per_inst_per_simd", "l2_read_bytes_per_launch", "l2_GBs",
                                      "hbm_bytes_per_launch", "hbm_GBs") if k in m}
    return r


class _OnlyTheLineOnStdout:
    """stdout carries ONE line, the JSON: whatever else a library writes there while the legs run (this image's RCCL prints a five-line banner on stdout when a
    communicator is created) goes to stderr.  File descriptor 1 points at stderr for the run; the line is written to the saved descriptor at the end."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def line(self, text):
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)  # (the C library's buffer: it holds the banner when stdout is a pipe)
        except Exception:
            pass
        os.write(self.saved, (text + "\n").encode())


def spawn_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here, the way the driver does (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1), and hand their exit code on.  Rank 0's JSON line goes straight to this process's stdout."""
    import socket
    import subprocess
    import torch
    share = os.environ.get("GSH_BENCH_SHARE_GPU") == "1"
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < a.gpus and not share:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {visible} GPU(s) visible on this node (one rank per GPU; GSH_BENCH_SHARE_GPU=1 is the self-test "
                         "that puts every rank on GPU 0, with GSH_RCCL_LIBRARY naming the stand-in collective library)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    the_line = _OnlyTheLineOnStdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): they must agree")
    import torch
    import gnss_sdr_amd
    from gnss_sdr_amd.tracking import CorrelatorBank
    from gnss_sdr_amd.codes import gps_l1_ca_code
    import oracle  # the checker (spot check below) and the CPU baseline only; nothing the GPU legs consume comes from it

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if os.environ.get("GSH_BENCH_SHARE_GPU") == "1":
        local = 0  # self-test of the N > 1 code path on a one-GPU box (with GSH_BENCH_BACKEND=gloo and GSH_RCCL_LIBRARY=the stand-in): every rank uses GPU 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from gnss_sdr_amd.sharding import ControlPlane, weak_channel_prn
    # "nccl" is RCCL on ROCm; only the barrier / MAX-reduce / id hand-over use it (gnss-sdr_amd/sharding.py)
    cp = ControlPlane(os.environ.get("GSH_BENCH_BACKEND", "nccl"), device=dev)
    assert (cp.rank, cp.world) == (rank, world)

    fs, n, C, E, T = a.fs, int(round(a.fs * 1e-3)), a.channels, a.epochs, a.taps
    block = (E + 2) * n                    # one stream block: E epochs + the run-in the per-channel window offsets need
    block += (-block) % 2
    NB, BPS = max(a.ring_blocks, 1), max(a.blocks_per_step, 1)
    grouped = world > 1 or os.environ.get("GSH_BENCH_FORCE_DIST") == "1"

    # ---- inputs, resident in HBM before the timed region
    if rank == 0:
        x0, dop, cph = make_stream_torch(torch, dev, block, fs)
    else:
        x0 = torch.zeros(block, dtype=torch.complex64, device=dev)
        dop, cph = np.zeros(0), np.zeros(0)
    cs = torch.cuda.Stream(device=dev)      # a real (non-null) stream for the engine's launches
    stream = cs.cuda_stream
    bank = CorrelatorBank(C, 1023, device=local)
    for c in range(C):
        bank.set_code(c, gps_l1_ca_code(weak_channel_prn(rank, C, c)))
    jobs, rows = build_jobs(C, E, n, fs, T, dop, cph, rank)
    bank.set_splits(1)
    G = ring = raw_src = None
    more_banks, more_streams = [], []
    if not grouped:
        # N = 1: NB copies of the block back to back; the steps cycle through them (643 MB: what misses L2 comes from HBM, not from the
        # 256 MiB Infinity Cache)
        x = torch.empty(NB * block, dtype=torch.complex64, device=dev)
        for k in range(NB):
            x[k * block:(k + 1) * block] = x0
        bank.set_stream_device(x.data_ptr(), NB * block, keepalive=x)
        bank.upload_jobs(jobs)
        # launches in flight: one more bank (its own job table and output rows) and stream per extra launch -- consecutive blocks are independent jobs of the
        # open-loop path, so block b + 1 may start while block b drains
        for _ in range(max(a.launches_in_flight, 1) - 1):
            b2 = CorrelatorBank(C, 1023, device=local)
            for c in range(C):
                b2.set_code(c, gps_l1_ca_code(weak_channel_prn(rank, C, c)))
            b2.set_splits(1)
            b2.set_stream_device(x.data_ptr(), NB * block, keepalive=x)
            b2.upload_jobs(jobs)
            more_banks.append(b2)
            more_streams.append(torch.cuda.Stream(device=dev))
    else:
        # N > 1 (or the self-test of that path): every block reaches the GPUs through the engine's stream group -- 8-bit items in, complex64
        # in every GPU's ring -- and the correlator bank reads the ring
        from gnss_sdr_amd.sample_stream import StreamGroup
        uid = cp.communicator_id()                      # 128 bytes of control plane (None for a world of one)
        G = StreamGroup.from_rank(local, rank, world, uid, 3 * block + 2, block // 2, os.environ.get("GSH_BENCH_DIST", "broadcast"))
        ring = G.ring(0)
        if rank == 0:
            raw_src = torch.view_as_real(x0).mul(30.0).round_().clamp_(-127, 127).to(torch.int8).reshape(-1).contiguous()
        torch.cuda.synchronize()
        first = G.push_device(raw_src.data_ptr() if raw_src is not None else None, block, "ibyte")
        G.wait()
        bank.set_stream_ring(ring)
        bank.upload_jobs(jobs)                          # window positions relative to the start of a block, staged once
        x = None

    state = {"blk": 0, "next_first": None}

    def step(k, in_flight=None):
        if not grouped:
            banks = [bank] + (more_banks if in_flight is None else more_banks[:in_flight - 1])
            streams = [stream] + [t.cuda_stream for t in more_streams]
            for j in range(BPS):
                b = (k * BPS + j) % len(banks)
                banks[b].set_sample_base(((k * BPS + j) % NB) * block)
                banks[b].launch(streams[b])
            return
        for j in range(BPS):
            # block b + 1 is queued for replication, then block b is correlated: the bank waits for the push that covers ITS windows only,
            # and the push waits only for launches that still read what it overwrites (ring of three blocks)
            if state["next_first"] is None:
                state["next_first"] = G.push_device(raw_src.data_ptr() if raw_src is not None else None, block, "ibyte")
            cur = state["next_first"]
            state["next_first"] = G.push_device(raw_src.data_ptr() if raw_src is not None else None, block, "ibyte")
            bank.set_sample_base(cur)
            bank.launch(stream)

    with torch.cuda.stream(cs):
        for k in range(max(a.settle_steps, 0)):   # set-up, not warm-up: bring the clocks out of their idle state with the same launches
            step(k)
        for k in range(a.warmup):
            step(a.settle_steps + k)
        torch.cuda.synchronize()
        cp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(a.settle_steps + a.warmup + k)
        torch.cuda.synchronize()
        if G is not None:
            G.wait()
        cp.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    reduce_max, reduce_sum = cp.reduce_max, cp.reduce_sum
    dt = reduce_max(dt)
    # the same steps with ONE launch at a time on one stream (what `value` was until round 6), over a fifth of the steps
    dt_single, steps_single = None, 0
    if not grouped and more_banks:
        steps_single = max(2, a.steps // 5)
        k0 = a.settle_steps + a.warmup + a.steps
        with torch.cuda.stream(cs):
            step(k0, in_flight=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps_single):
                step(k0 + 1 + k, in_flight=1)
            torch.cuda.synchronize()
            dt_single = time.perf_counter() - t0
        last_launch = (k0 + 1 + steps_single) * BPS - 1
    else:
        last_launch = (max(a.settle_steps, 0) + a.warmup + a.steps) * BPS - 1

    # ---- roofline of the dominant kernel: HIP events on the launch stream, inputs resident; taken straight after the timed region,
    # before the host-side spot check lets the GPU fall idle again
    bank.synchronize()
    k_ms = bank.time_launches(100)

    # ---- spot-check against the oracle (not timed): a few jobs of the last launch
    out = bank.read_outputs()
    from helpers import oracle_job, scale_err
    if not grouped:
        # time_launches() re-ran the last launch: the block the final step ended on
        base_last = last_launch % NB * block
        xh = x[base_last:base_last + block].cpu().numpy()
    else:
        # the ring holds the 8-bit block converted back to float: that is what the kernel correlated.  Only rank 0 made the block; every other rank checks
        # what REACHED ITS RING through the group (its own copy read back from its own HBM) -- a wrong chunk offset anywhere shows here
        lo_r, hi_r = ring.range()
        xh = np.concatenate([ring.read(hi_r - block + k, min(1 << 20, block - k)) for k in range(0, block, 1 << 20)])
        if rank == 0:
            want = raw_src.to(torch.float32).reshape(-1, 2).cpu().numpy()
            if not np.array_equal(xh.view(np.float32).reshape(-1, 2), want):
                raise SystemExit("bench: rank 0's ring does not hold the block it pushed")
    worst = 0.0
    for j in (0, 1, C + 3, len(rows) - 1):
        o32, t64, sabs = oracle_job(oracle.ca_code(weak_channel_prn(rank, C, rows[j]["code_slot"])), xh, rows[j])
        err = scale_err(out[j, :T], t64, sabs)
        if not np.all(err <= 1e-6):
            raise SystemExit(f"bench: rank {rank}: GPU result of job {j} disagrees with the oracle: {out[j, :T]} vs {t64}")
        worst = max(worst, float(np.max(err)))
    for b2 in more_banks:
        # the other banks computed too: three jobs of their last launch against the oracle (the ring's blocks are copies of one block, so whichever block
        # a bank ended on, xh holds its samples)
        out2 = b2.read_outputs()
        for j in (0, C + 3, len(rows) - 1):
            o32, t64, sabs = oracle_job(oracle.ca_code(weak_channel_prn(rank, C, rows[j]["code_slot"])), xh, rows[j])
            err = scale_err(out2[j, :T], t64, sabs)
            if not np.all(err <= 1e-6):
                raise SystemExit(f"bench: rank {rank}: job {j} of a second bank disagrees with the oracle: {out2[j, :T]} vs {t64}")
            worst = max(worst, float(np.max(err)))
    if grouped:
        # every rank's ring must hold the same block: a checksum of checksums over the ranks
        crc = float(int(np.frombuffer(xh.tobytes(), dtype=np.uint32).sum(dtype=np.uint64)) % (1 << 40))
        if not cp.same_everywhere(crc):
            raise SystemExit(f"bench: rank {rank}: the replicated block differs between the ranks")
    spot = {"ranks_checked": int(round(reduce_sum(1.0))), "jobs_per_rank": 4, "worst_err": reduce_max(worst), "bar": 1e-6,
            "what": "|gpu - float64 truth| / sum|x| of 4 jobs of the last launch on EVERY rank" + (", each against its own ring's copy of the replicated block" if grouped else "")}

    sharded = None
    if grouped:
        # the closed loop and the acquisition on the replicated stream (every rank takes part; see sharded_legs)
        sharded = sharded_legs(torch, cp, dev, local, rank, world, ring, block, fs, n, C)
    if rank == 0:
        pmc = load_pmc()
        total_corr = float(C) * T * E * BPS * a.steps * world
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
            "config": {"workload": f"GPS L1 C/A tracking, {C} channels/GPU x {E} epochs/block x {BPS} blocks/step, fs={fs / 1e6:g} Msps, N={n}, {T}-tap E/P/L, open-loop, "
                                   + (f"{1 + len(more_banks)} launches in flight (blocks alternate over that many banks and streams); " if more_banks else "one launch at a time; ")
                                   + ("the IF stream is RESIDENT in HBM before the timed region (host-to-device transfer excluded: pcie_inclusive is the end-to-end figure)"
                                      if not grouped else "every 8-bit block enters GPU 0 from device memory and is replicated by the engine inside the timed region"),
                       "channels_per_gpu": C, "epochs_per_block": E, "blocks_per_step": BPS, "samples_per_epoch": n, "taps": T,
                       "stream_resident_bytes": int(8 * NB * block) if not grouped else int(8 * (3 * block + 2)),
                       "timed_region_s": dt,
                       "parallelism": f"channels sharded over {world} GPU(s)" + (f", every 8-bit stream block replicated by the engine's RCCL stream group "
                                      f"({os.environ.get('GSH_BENCH_DIST', 'broadcast')}), converted into every GPU's ring, overlapped with the correlation" if grouped else "")},
            "roofline": tracking_roofline(C, E, T, n, k_ms, pmc),
            "kernel_only_value": float(C) * T * E / (k_ms * 1e-3),
            # `value` = blocks alternating over launches_in_flight banks and streams; value_single_stream = one launch after the other on one stream (what
            # `value` was until round 6), timed straight after over a fifth of the steps; roofline.kernel_ms = one launch alone (HIP events, one stream)
            "launches_in_flight": 1 + len(more_banks),
            "value_single_stream": (float(C) * T * E * BPS * steps_single / dt_single) if dt_single else None,
            "ms_per_step_single_stream": (dt_single / steps_single * 1e3) if dt_single else None,
            # the stream group under the launches (N > 1: RCCL over xGMI inside the engine; N = 1: no group, the block is resident)
            "rccl_ranks": G.rccl_info()["ranks"] if G is not None else 0,
            "stream_group_mode": os.environ.get("GSH_BENCH_DIST", "broadcast") if grouped else None,
            "spot_check": spot,
        }
        if G is not None and res["rccl_ranks"] > 0:
            from gnss_sdr_amd.sample_stream import StreamGroup
            res["rccl_library"] = StreamGroup.library()
            res["rccl_library_is_test_stub"] = G.rccl_info()["version"] == 99999   # tests/host/fake_rccl.cc: a functional self-test, its rates mean nothing
            res["rccl_calls"] = G.rccl_info()["collectives"]
        # the figures a reader needs first, as top-level scalars (the nested objects explain them)
        rf = res["roofline"]
        res["traffic_source"] = rf.get("traffic_source")
        res["contract_hbm_rate_over_peak"] = rf["contract_hbm"]["rate_over_peak"]   # SURVEY 8(d) bytes / time / 8 TB/s: a RATE (> 1: 32 channels share one stream)
        res["hbm_unique_frac"] = rf["hbm_unique"]["frac"]                            # bytes that must leave HBM once per launch / time / 8 TB/s: the utilisation
        res["valu_issue_frac"] = rf.get("valu_issue_frac")                           # what binds the kernel (static counters)
        try:
            import ctypes
            gbs = ctypes.c_double(0.0)
            from gnss_sdr_amd import _lib as _gl
            _gl.check(_gl.load().gsh_probe_read_bandwidth(local, 2 << 30, 8, ctypes.byref(gbs)))
            res["hbm_read_probe_GBs"] = gbs.value        # measured streaming read of 2 GiB (> Infinity Cache) with 16-byte loads, this device, this run
            res["hbm_read_probe_frac_of_nominal"] = gbs.value / HBM_PEAK_GBS
            res["contract_hbm_rate_over_measured"] = rf["contract_hbm"]["achieved"] / gbs.value
            res["hbm_unique_frac_of_measured"] = rf["hbm_unique"]["achieved_GBs"] / gbs.value
        except Exception as e:
            res["hbm_read_probe_GBs"] = None
            res["hbm_read_probe_error"] = str(e)
        if sharded is not None:
            res.update(sharded)
        # ---- the other GPU legs first, while the device is still at its working clocks (the CPU legs below leave it idle for ~40 s; what runs after
        # them starts from the idle power state however long its own warm-up is)
        if world == 1 and not a.no_acq and not grouped:
            try:
                res["acquisition"] = acquisition_metric(torch, local, x0[:n].contiguous(), fs, pmc)
            except Exception as e:
                res["acquisition"] = {"error": str(e)}
            try:
                res["closed_loop"] = closed_loop_metric(local, x0, block, fs, n, dop, cph, channels=C, epochs=min(E - 2, 200))
                # ... with the lock detectors on (what the tracking adapters run), launched and as one live residency
                res["closed_loop_lock_detectors"] = closed_loop_metric(local, x0, block, fs, n, dop, cph, channels=C, epochs=min(E - 2, 200), lock_detectors=True, live=True)
                # one compute unit per channel: 32 channels use an eighth of the chip, 256 (BASELINE config 5's channel count) fill it
                res["closed_loop_256ch"] = closed_loop_metric(local, x0, block, fs, n, dop, cph, channels=256, epochs=min(E - 2, 200))
            except Exception as e:
                res["closed_loop"] = {"error": str(e)}
            try:
                res["closed_loop_config4"] = closed_loop_config4_metric(torch, local)
            except Exception as e:
                res["closed_loop_config4"] = {"error": str(e)}
            # the same two shapes with cooperating work-groups (gsh_trk_set_split: an option -- sums in another order, records equal to rounding): where the
            # channels leave most of the chip idle and a window is long, several compute units share every window of a channel
            try:
                res["closed_loop_cooperating"] = {
                    "config2_2_work_groups": closed_loop_metric(local, x0, block, fs, n, dop, cph, channels=C, epochs=min(E - 2, 200), lock_detectors=True, split=2),
                    "config4_4_work_groups": closed_loop_config4_metric(torch, local, split=4)}
            except Exception as e:
                res["closed_loop_cooperating"] = {"error": str(e)}
            if not a.no_other_configs:
                try:
                    res["other_configs"] = other_configs_metric(local)
                except Exception as e:
                    res["other_configs"] = {"error": str(e)}
            try:
                res["pcie_inclusive"] = pcie_inclusive_metric(torch, local, x0, jobs, C, E, T, block)
            except Exception as e:
                res["pcie_inclusive"] = {"error": str(e)}
            try:
                res["rccl_one_rank"] = rccl_one_rank_metric(torch, local, x0, block)
            except Exception as e:
                res["rccl_one_rank"] = {"error": str(e)}
        # ---- then the legs with a CPU part: the drop-in seam (32 block threads + 32 reference blocks as the checker) and the CPU baseline
        if world == 1 and not a.no_dropin and not grouped:
            try:
                # the reference's cadence (one code period per general_work call, trk.cc:1898-2001) and 20 periods per call, each over a steady window of seconds
                res["dropin"] = dropin_metric(C, fs, 4000000, 20, a.dropin_seconds)
                res["dropin"]["one_period_per_call"] = dropin_metric(C, fs, 4000000, 1, a.dropin_seconds)
            except Exception as e:
                res["dropin"] = {"error": str(e)}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(C, n, fs, T, a.cpu_seconds)
        res["summary"] = bench_summary(res)
        the_line.line(json.dumps(res))
    bank.close()
    for b2 in more_banks:
        b2.close()
    if G is not None:
        G.close()
    cp.close()


if __name__ == "__main__":
    main()
