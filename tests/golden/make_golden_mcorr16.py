#!/usr/bin/env python3
"""Mint tests/golden/mcorr16.npz from the REFERENCE ITSELF: Cpu_Multicorrelator_16sc over the generic protokernels
(cpu_multicorrelator_16sc.cc and K/volk_gnsssdr_16ic_* compiled from /root/reference by oracle/Makefile into oracle/_ref).
Run in the build container only:

    python tests/golden/make_golden_mcorr16.py

Every case stores its inputs (int16 samples and code, float32 parameters) next to the reference's int16 outputs.  The shapes cover the
volk QA's vector length (8111, kernel_tests.h:85), the phasor's renormalisation points (n around 256), windows shorter than a wave, sums that
stay far from saturation, sums that saturate early and stay there, sums that saturate and come back, chip indices that start negative, and the
BASELINE window (25 000 samples, 1023-chip code, E/P/L).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def cases():
    rng = np.random.default_rng(0x16C0DE)
    ca = np.stack([oracle.ca_code(5), np.zeros(1023, np.float32)], -1).astype(np.int16)
    out = []

    def add(name, n, amp, code, shifts, rem_carr, step, rem_code, cstep, shape="uniform"):
        if shape == "uniform":
            x = rng.integers(-amp, amp + 1, size=(n, 2)).astype(np.int16)
        elif shape == "bias":   # a constant on top of noise: the sums run into a bound and stay
            x = (rng.integers(-3, 4, size=(n, 2)) + amp).astype(np.int16)
        else:                   # "swing": a bias over the first sixth, zero-mean noise afterwards: the sums hit a bound, then wander off it again
            on = np.where(np.arange(n) < n // 6, 1, 0)[:, None]
            x = (rng.integers(-40, 41, size=(n, 2)) + amp * on).astype(np.int16)
        out.append(dict(name=name, x=x, code=code, shifts=np.asarray(shifts, np.float32),
                        par=np.asarray([rem_carr, step, rem_code, cstep], np.float32)))

    add("one_sample", 1, 100, ca, [-0.5, 0.0, 0.5], 1.0, 0.1, 0.2, 0.04092)
    add("short_63", 63, 50, ca, [0.0], 0.3, -0.02, 0.9, 0.2557)
    add("wave_64", 64, 50, ca, [-0.5, 0.0, 0.5], 2.3, 0.05, 0.0, 0.2557)
    add("renorm_255", 255, 300, ca, [-0.5, 0.0, 0.5], 5.9, 0.31, 0.5, 0.2557)
    add("renorm_256", 256, 300, ca, [-0.5, 0.0, 0.5], 5.9, 0.31, 0.5, 0.2557)
    add("renorm_257", 257, 300, ca, [-0.5, 0.0, 0.5], 5.9, 0.31, 0.5, 0.2557)
    add("qa_8111", 8111, 30, ca, [-0.6, -0.3, 0.0, 0.3, 0.6], 0.7, 0.0212, 0.45, 0.1263)
    add("baseline_25000", 25000, 40, ca, [-0.5, 0.0, 0.5], 4.1, 0.00118, 0.77, 0.04092)
    add("full_scale", 4000, 32767, ca, [-0.5, 0.0, 0.5], 0.78539819, 0.003, 0.1, 0.2557)   # rotated magnitudes beyond int16: the cast wraps
    add("saturate_stay", 5000, 900, ca, [-0.5, 0.0, 0.5], 0.0, 0.0, 0.0, 0.0, shape="bias")  # step 0: one chip, one phase: sums hit a bound at once
    add("saturate_return", 6000, 700, ca, [-0.5, 0.0, 0.5], 0.0, 0.0, 0.0, 0.0, shape="swing")
    cplx = rng.integers(-9, 10, size=(10230, 2)).astype(np.int16)
    add("complex_code_10230", 12000, 25, cplx, [-1.0, 0.0, 1.0], 3.3, -0.0471, 7.25, 0.4092)
    add("negative_indices", 3000, 60, ca, [-2.5, 0.0, 0.5], 1.9, 0.02, 2100.75, 0.2557)      # rem_code beyond two code lengths
    add("negative_step", 3000, 60, ca, [-0.5, 0.0, 0.5], 1.9, 0.02, 0.5, -0.2557)
    add("eight_taps", 2048, 45, ca, np.linspace(-1.75, 1.75, 8), 0.01, 0.2, 0.3, 0.2557)
    return out


def main():
    if oracle.ref() is None or not hasattr(oracle.ref(), "ref_mcorr16_run"):
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    store = {}
    names = []
    for c in cases():
        p = [float(v) for v in c["par"]]
        want = oracle.ref_mcorr16(c["code"], c["shifts"], c["x"], p[0], p[1], p[2], p[3])
        names.append(c["name"])
        for k in ("x", "code", "shifts", "par"):
            store[f"{c['name']}__{k}"] = c[k]
        store[f"{c['name']}__out"] = want
        print(f"{c['name']:22s} n {len(c['x']):6d} -> {want.tolist()}")
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "mcorr16.npz"), **store)


if __name__ == "__main__":
    main()
