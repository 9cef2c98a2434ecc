"""CPU tests: the oracle (oracle/gnss_oracle.c) against the committed golden vectors (minted from the reference
itself by tests/golden/make_golden.py) and, where oracle/_ref is present, against the live reference build.

Integer / code outputs: exact.  float32 outputs of the reference-order restatement: exact (bit-for-bit) -- the
restatement performs the same IEEE operations in the same order as the reference's _generic protokernels.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def codes():
    return dict(np.load(os.path.join(G, "codes.npz")))


def test_ca_codes_match_reference(codes):
    for prn in list(range(1, 33)) + [120, 131, 138]:
        assert np.array_equal(oracle.ca_code(prn).astype(np.int8), codes[f"ca_{prn}"]), prn
    assert np.array_equal(oracle.ca_code(7, 13).astype(np.int8), codes["ca_7_shift13"])
    for fs in (4000000, 25000000):
        s = oracle.ca_code_complex_sampled(5, fs)
        assert np.all(s.real == 0)
        assert np.array_equal(s.imag.astype(np.int8), codes[f"ca_sampled_5_{fs}"])
    with pytest.raises(ValueError):
        oracle.ca_code(0)
    with pytest.raises(ValueError):
        oracle.ca_code(139)


def test_ca_code_properties():
    # balance and the three-valued cross-correlation of Gold codes: independent of any fixture
    c1, c2 = oracle.ca_code(1).astype(np.int32), oracle.ca_code(2).astype(np.int32)
    assert abs(int(c1.sum())) == 1
    auto = np.array([np.dot(c1, np.roll(c1, k)) for k in range(1, 1023)])
    cross = np.array([np.dot(c1, np.roll(c2, k)) for k in range(1023)])
    assert set(np.unique(auto)) <= {-65, -1, 63}
    assert set(np.unique(cross)) <= {-65, -1, 63}


def test_mcorr_matches_golden(codes):
    g = dict(np.load(os.path.join(G, "mcorr.npz")))
    names = sorted({k[:-4] for k in g if k.endswith("_out")})
    assert len(names) >= 6
    for name in names:
        ck, mode = g[name + "_meta"]
        mode = int(mode)
        code = codes[ck].astype(np.float32)
        rc, ps, pr, rcode, cs, cr = [float(v) for v in g[name + "_par"]]
        out = oracle.mcorr(code, g[name + "_shifts"], g[name + "_x"], rem_carr=rc, phase_step=ps, rem_code=rcode,
                           code_step=cs, phase_rate_step=pr, code_rate_step=cr, high_dyn=mode)
        assert np.array_equal(out.view(np.float32), g[name + "_out"].view(np.float32)), name
        # and the float64 truth sits where the reference's own float32 error budget says it should
        t64, sabs = oracle.mcorr_f64(code, g[name + "_shifts"], g[name + "_x"], rem_carr=rc, phase_step=ps, rem_code=rcode,
                                     code_step=cs, phase_rate_step=pr, code_rate_step=cr, high_dyn=mode)
        tol = 2e-6 if mode != 1 else 2e-5  # the hd rotator never renormalises phase_doppler
        assert np.all(np.abs(out - t64) / sabs < tol), (name, np.abs(out - t64) / sabs)


def test_small_kernels_match_golden():
    g = dict(np.load(os.path.join(G, "small_kernels.npz")))
    out = np.empty(8000, np.float32)
    ph = C.c_float(0.0)
    oracle.lib().oracle_sincos(out, float(g["sincos_step"][0]), C.byref(ph), 4000)
    assert np.array_equal(out, g["sincos_out"])
    assert np.float32(ph.value) == g["sincos_final_phase"][0]
    t = np.zeros(1, np.uint32)
    oracle.lib().oracle_index_max(t, g["imax_in"], len(g["imax_in"]))
    assert t[0] == g["imax_out"][0] == 1234


def test_code_indices_wrap_and_monotone():
    # negative starts wrap like the reference (K/..resampler..:75-76), indices are monotone mod L
    idx = oracle.code_indices(5000, [-1.5, 0.0, 1.5], rem_code=0.7, code_step=0.26, code_len=1023)
    assert idx.min() >= 0 and idx.max() < 1023
    assert idx[0, 0] == 1023 - 3 and idx[1, 0] == 1022 and idx[2, 0] == 0
    d = np.diff(idx.astype(np.int64), axis=1) % 1023
    assert set(np.unique(d)) <= {0, 1}


@pytest.mark.skipif(oracle.ref() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_live_reference():
    rng = np.random.default_rng(1)
    code = oracle.ca_code(3)
    for n, mode in [(4000, 0), (25000, 0), (8111, 0), (4000, 1), (25000, 1), (70000, 1), (8192, 2), (1, 0), (17, 0)]:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        sh = np.array([-0.5, 0.0, 0.5], np.float32)
        kw = dict(rem_carr=float(rng.uniform(0, 6.28)), phase_step=float(rng.uniform(-0.01, 0.01)),
                  rem_code=float(rng.uniform(0, 1)), code_step=float(rng.uniform(0.03, 0.3)),
                  phase_rate_step=1e-9 if mode == 1 else 0.0, code_rate_step=1e-12 if mode else 0.0, high_dyn=mode)
        if mode and n < 64:
            continue
        a = oracle.mcorr(code, sh, x, **kw)
        b = oracle.ref_mcorr(code, sh, x, **kw)
        assert np.array_equal(a.view(np.float32), b.view(np.float32)), (n, mode)
    # the reference's own SIMD protokernel is ~1e-5 away from its generic one (documented tolerance context)
    x = (rng.standard_normal(25000) + 1j * rng.standard_normal(25000)).astype(np.complex64)
    kw = dict(rem_carr=1.0, phase_step=0.002, rem_code=0.3, code_step=0.04092)
    gen = oracle.ref_mcorr(code, [-0.5, 0, 0.5], x, **kw)
    simd = oracle.ref_mcorr(code, [-0.5, 0, 0.5], x, simd=True, **kw)
    assert np.all(np.abs(gen - simd) / np.abs(gen) < 1e-3)  # the reference's QA bound, kernel_tests.h:41,88-89


def test_e1_l5_fixture_properties():
    """tests/golden/codes_e1_l5.npz: shapes, +-1 values, sinBOC(1,1) structure (each E1 chip is the pair {+c, -c},
    galileo_e1_signal_replica.cc:98-108) and the small per-PRN samples stored in codes.npz by the same script."""
    from helpers import golden_e1_l5_codes
    import os
    g = golden_e1_l5_codes()
    assert g["e1b"].shape == (50, 8184) and g["e1c"].shape == (50, 8184)
    assert g["l5i"].shape == (32, 10230) and g["l5q"].shape == (32, 10230)
    for k, v in g.items():
        assert np.all(np.abs(v) == 1.0), k
    for k in ("e1b", "e1c"):
        assert np.array_equal(g[k][:, 0::2], -g[k][:, 1::2]), k
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "codes.npz"))
    for prn in (1, 11, 50):
        assert np.array_equal(z[f"e1_1B_{prn}"].astype(np.float32), g["e1b"][prn - 1])
        assert np.array_equal(z[f"e1_1C_{prn}"].astype(np.float32), g["e1c"][prn - 1])
    for prn in (1, 32):
        assert np.array_equal(z[f"l5i_{prn}"].astype(np.float32), g["l5i"][prn - 1])
        assert np.array_equal(z[f"l5q_{prn}"].astype(np.float32), g["l5q"][prn - 1])
    # distinct PRNs are nearly orthogonal over a period
    c = g["l5i"] @ g["l5i"].T / 10230.0
    assert np.all(np.abs(c - np.eye(32)) < 0.05)


@pytest.mark.skipif(oracle.ref() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_e1_l5_fixture_equals_live_reference():
    from helpers import golden_e1_l5_codes
    g = golden_e1_l5_codes()
    R = oracle.ref()
    b = np.empty(8184, np.float32)
    for prn in (2, 25, 49):
        R.ref_galileo_e1_code_gen_sinboc11_float(b, b"1B", prn)
        assert np.array_equal(b, g["e1b"][prn - 1])
        R.ref_galileo_e1_code_gen_sinboc11_float(b, b"1C", prn)
        assert np.array_equal(b, g["e1c"][prn - 1])
    b = np.empty(10230, np.float32)
    for prn in (3, 17, 31):
        R.ref_gps_l5i_code_gen_float(b, prn)
        assert np.array_equal(b, g["l5i"][prn - 1])
        R.ref_gps_l5q_code_gen_float(b, prn)
        assert np.array_equal(b, g["l5q"][prn - 1])
