#!/usr/bin/env python3
"""Mint golden fixtures for the tracking-state pieces of SURVEY 8f-2 from the REFERENCE ITSELF (oracle/_ref: lock_detectors.cc,
exponential_smoother.cc, bit_synchronizer.cc compiled from /root/reference by oracle/Makefile).  Build container only:

    python tests/golden/make_golden_trackstate.py          -> tests/golden/trackstate.npz

Every *_out array is an output of reference code for the seeded input stored next to it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def main():
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_bit_sync_run"):
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    rng = np.random.default_rng(0x5EED0F2)
    g = {}
    # ---- cn0_m2m4_estimator / carrier_lock_detector on prompt buffers of several C/N0
    amps = [0.0, 0.3, 1.0, 5.0, 40.0, 1000.0]
    prompts = np.stack([(a * rng.choice([-1.0, 1.0], 20) + rng.standard_normal(20) + 1j * rng.standard_normal(20)).astype(np.complex64) for a in amps for _ in range(4)])
    cn0 = np.zeros((len(prompts), 3), np.float32)
    lock = np.zeros((len(prompts), 2), np.float32)
    for i, p in enumerate(prompts):
        pf = np.ascontiguousarray(p).view(np.float32)
        for j, t in enumerate((0.001, 0.004, 0.02)):
            cn0[i, j] = R.ref_cn0_m2m4_estimator(pf, 20, t)
        lock[i, 0] = R.ref_carrier_lock_detector(pf, 20)
        lock[i, 1] = R.ref_carrier_lock_detector(pf, 1)
    g["lock_prompts"], g["lock_cn0_out"], g["lock_detector_out"] = prompts, cn0, lock
    # ---- Exponential_Smoother: (alpha, samples_for_initialization, min_value, offset) as trk.cc:680-692 configures its two instances
    cfgs = np.array([(0.002, 200, 25.0, 12.0), (0.002, 50, 25.0, 12.0), (0.002, 25, -1.0, 0.0), (0.5, 1, -1.0, 0.0)], np.float32)
    raws, outs = [], []
    for (alpha, n_init, mn, off) in cfgs:
        for base in (10.0, 36.9, 37.1, 45.0, 0.9):
            raw = (base + rng.standard_normal(600)).astype(np.float32)
            raw[300:340] -= 30.0
            out = np.zeros_like(raw)
            R.ref_smoother_run(float(alpha), int(n_init), float(mn), float(off), raw, len(raw), out)
            raws.append(raw)
            outs.append(out)
    g["smoother_cfg"], g["smoother_raw"], g["smoother_out"] = cfgs, np.stack(raws), np.stack(outs)
    # ---- HistogramBitSynchronizer
    bs_cfg, bs_p, bs_ok, bs_ev, bs_un = [], [], [], [], []
    for trial in range(12):
        bins = [20, 20, 10, 4][trial % 4]
        n = 600
        offset = int(rng.integers(0, bins))
        bits = rng.choice([-1.0, 1.0], n // bins + 2)
        amp = [30.0, 6.0, 2.0][trial % 3]
        sym = bits[(np.arange(n) + offset) // bins]
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi)) if trial % 2 else 1.0
        p = ((amp * sym + rng.standard_normal(n) + 1j * rng.standard_normal(n)) * ph).astype(np.complex64)
        ok = (rng.uniform(size=n) > (0.1 if trial % 5 == 0 else 0.0)).astype(np.int32)
        cfg = (bins, [10, 5][trial % 2], [3, 5][(trial // 2) % 2], [0.6, 0.4][(trial // 3) % 2], [0.0, 3.0][(trial // 4) % 2], int(trial % 2 == 0))
        ev, ph_o, un = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        R.ref_bit_sync_run(cfg[0], 1, cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], np.ascontiguousarray(p).view(np.float32), ok, n, ev, ph_o, un)
        bs_cfg.append(cfg)
        bs_p.append(p)
        bs_ok.append(ok)
        bs_ev.append(ev)
        bs_un.append(un)
    g["bitsync_cfg"] = np.array(bs_cfg, np.float64)
    g["bitsync_prompts"], g["bitsync_ok"], g["bitsync_event_out"], g["bitsync_until_edge_out"] = np.stack(bs_p), np.stack(bs_ok), np.stack(bs_ev), np.stack(bs_un)
    path = os.path.join(HERE, "trackstate.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
