#!/usr/bin/env python3
"""Mint the golden fixtures from the REFERENCE ITSELF (oracle/_ref = /root/reference sources compiled by
oracle/Makefile).  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  Everything stored is an output of reference code (PRN generators,
Cpu_Multicorrelator_Real_Codes over the _generic protokernels, volk_gnsssdr_s32f_sincos_32fc_generic,
volk_gnsssdr_32f_index_max_32u_generic) for the seeded inputs stored next to it.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def main():
    R = oracle.ref()
    if R is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")

    # ---- PRN codes ------------------------------------------------------------------------------------
    ca = {}
    for prn in list(range(1, 33)) + [120, 131, 138]:
        b = np.empty(1023, np.float32)
        R.ref_gps_l1_ca_code_gen_float(b, prn, 0)
        ca[f"ca_{prn}"] = b.astype(np.int8)
    b = np.empty(1023, np.float32)
    R.ref_gps_l1_ca_code_gen_float(b, 7, 13)
    ca["ca_7_shift13"] = b.astype(np.int8)
    for fs in (4000000, 25000000):
        n = fs // 1000
        s = np.zeros(2 * n, np.float32)
        R.ref_gps_l1_ca_code_gen_complex_sampled(s, n, 5, fs, 0)
        assert np.all(s[0::2] == 0)
        ca[f"ca_sampled_5_{fs}"] = s[1::2].astype(np.int8)  # code sits in the imaginary part
    e1 = {}
    for sig in ("1B", "1C"):
        for prn in (1, 11, 50):
            b = np.empty(2 * 4092, np.float32)
            R.ref_galileo_e1_code_gen_sinboc11_float(b, sig.encode(), prn)
            e1[f"e1_{sig}_{prn}"] = b.astype(np.int8)
    l5 = {}
    for prn in (1, 32):
        b = np.empty(10230, np.float32)
        R.ref_gps_l5i_code_gen_float(b, prn)
        l5[f"l5i_{prn}"] = b.astype(np.int8)
        R.ref_gps_l5q_code_gen_float(b, prn)
        l5[f"l5q_{prn}"] = b.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "codes.npz"), **ca, **e1, **l5)

    # ---- full tracking-replica sets for BASELINE configs 4 and 5 (Galileo E1 B/C sinBOC(1,1) at 2 samples per chip,
    # trk.cc:289,837-844; GPS L5 I/Q, trk.cc:903-915), bit-packed: bit = 1 where the reference's float chip is +1
    full = {}
    e1b = np.empty((50, 8184), np.int8)
    e1c = np.empty((50, 8184), np.int8)
    b = np.empty(8184, np.float32)
    for prn in range(1, 51):
        R.ref_galileo_e1_code_gen_sinboc11_float(b, b"1B", prn)
        assert np.all(np.abs(b) == 1.0)
        e1b[prn - 1] = b
        R.ref_galileo_e1_code_gen_sinboc11_float(b, b"1C", prn)
        e1c[prn - 1] = b
    full["e1b_sinboc11"] = np.packbits(e1b > 0, axis=1)
    full["e1c_sinboc11"] = np.packbits(e1c > 0, axis=1)
    l5i = np.empty((32, 10230), np.int8)
    l5q = np.empty((32, 10230), np.int8)
    b = np.empty(10230, np.float32)
    for prn in range(1, 33):
        R.ref_gps_l5i_code_gen_float(b, prn)
        assert np.all(np.abs(b) == 1.0)
        l5i[prn - 1] = b
        R.ref_gps_l5q_code_gen_float(b, prn)
        l5q[prn - 1] = b
    full["l5i"] = np.packbits(l5i > 0, axis=1)
    full["l5q"] = np.packbits(l5q > 0, axis=1)
    np.savez_compressed(os.path.join(HERE, "codes_e1_l5.npz"), **full)

    # ---- loop closure: discriminators, loop filters (the reference's own objects) ----------------------------
    if hasattr(R, "ref_fll_diff_atan"):
        rl = np.random.default_rng(20260923)
        v = (rl.standard_normal((64, 8)) * 1000.0).astype(np.float32)
        v[0] = 0.0
        v[1, 0] = 0.0
        outs = np.array([[R.ref_pll_cloop_two_quadrant_atan(*map(float, r[:2])), R.ref_fll_diff_atan(*map(float, r[:4]), 0.0, 0.001),
                          R.ref_dll_nc_e_minus_l_normalized(*map(float, r[:4]), 0.5, 1.0, 1.0), R.ref_dll_nc_vemlp_normalized(*map(float, r))]
                         for r in v])
        loop = {"disc_inputs": v, "disc_outputs": outs}
        lf_in = rl.standard_normal(100).astype(np.float32)
        loop["lf_in"] = lf_in
        for order in (1, 2, 3):
            for last in (0, 1):
                o = np.zeros(len(lf_in), np.float32)
                R.ref_loop_filter_run(0.001, 2.0, order, last, 0.0, lf_in, o, len(lf_in))
                loop[f"lf_out_{order}_{last}"] = o
        fp_fll = (rl.standard_normal(100) * 3).astype(np.float32)
        fp_pll = (rl.standard_normal(100) * 0.05).astype(np.float32)
        loop["fp_fll"], loop["fp_pll"] = fp_fll, fp_pll
        for order in (2, 3):
            o = np.zeros(100, np.float32)
            R.ref_fll_pll_filter_run(35.0, 35.0, order, 1500.0, fp_fll, fp_pll, 0.001, o, 100)
            loop[f"fp_out_{order}"] = o
        np.savez_compressed(os.path.join(HERE, "loop.npz"), **loop)

    # ---- multicorrelator known answers ----------------------------------------------------------------
    rng = np.random.default_rng(20260922)
    cases = {}
    specs = [
        # name, code key, n, shifts, rem_carr, phase_step, phase_rate, rem_code, code_step, code_rate, mode
        ("std_4000", "ca_1", 4000, [-0.5, 0.0, 0.5], 0.7, 2 * np.pi * 1680.0 / 4e6, 0.0, 0.31, 1.023e6 / 4e6, 0.0, 0),
        ("std_8111_5tap", "ca_10", 8111, [-0.5, -0.15, 0.0, 0.15, 0.5], 5.9, 0.1, 0.0, 0.4, 0.3, 0.0, 0),
        ("std_25000", "ca_32", 25000, [-0.5, 0.0, 0.5], 3.3, 2 * np.pi * -4321.0 / 25e6, 0.0, 0.93, 1.023e6 * (1 - 4321.0 / 1575.42e6) / 25e6, 0.0, 0),
        ("hd7_8192", "ca_1", 8192, [-0.5, 0.0, 0.5], 0.0, 0.1, 1e-9, 0.4, 0.3, 1e-5, 1),
        ("hd6_8192", "ca_1", 8192, [-0.5, 0.0, 0.5], 0.0, 0.1, 0.0, 0.4, 0.3, 1e-5, 2),
        ("e1_veml_16000", "e1_1B_11", 16000, [-1.0, -0.3, 0.0, 0.3, 1.0], 1.1, 2 * np.pi * 900.0 / 4e6, 0.0, 0.5, 2 * 1.023e6 / 4e6, 0.0, 0),
    ]
    allcodes = {**ca, **e1, **l5}
    for name, ck, n, sh, rc, ps, pr, rcode, cs, cr, mode in specs:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        code = allcodes[ck].astype(np.float32)
        out = oracle.ref_mcorr(code, sh, x, rem_carr=rc, phase_step=ps, rem_code=rcode, code_step=cs,
                               phase_rate_step=pr, code_rate_step=cr, high_dyn=mode)
        cases[name + "_x"] = x
        cases[name + "_out"] = out
        cases[name + "_par"] = np.array([rc, ps, pr, rcode, cs, cr], np.float32)
        cases[name + "_shifts"] = np.array(sh, np.float32)
        cases[name + "_meta"] = np.array([ck, str(mode)])
    np.savez_compressed(os.path.join(HERE, "mcorr.npz"), **cases)

    # ---- small acquisition-side kernels ---------------------------------------------------------------
    small = {}
    out = np.empty(2 * 4000, np.float32)
    ph = C.c_float(0.0)
    step = np.float32(-(np.float32(2 * np.pi) * np.float32(1750.0) / np.float32(4e6)))
    R.ref_sincos_generic(out, float(step), C.byref(ph), 4000)
    small["sincos_step"] = np.array([step], np.float32)
    small["sincos_out"] = out.copy()
    small["sincos_final_phase"] = np.array([ph.value], np.float32)
    v = rng.random(5000).astype(np.float32)
    v[1234] = v[4321] = 2.0  # tie: the lowest index must win
    t = np.zeros(1, np.uint32)
    R.ref_index_max_generic(t, v, len(v))
    small["imax_in"] = v
    small["imax_out"] = t.copy()
    np.savez_compressed(os.path.join(HERE, "small_kernels.npz"), **small)
    for f in ("codes.npz", "mcorr.npz", "small_kernels.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
