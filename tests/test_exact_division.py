"""gnss-sdr_amd/csrc/exact_division.h -- the closed loop's divisions by launch constants (sampling rate, carrier frequency, 2 pi) and its fmod(phase, 2 pi) --
compiled for the host and compared, bit for bit, with the machine's own division and fmod: random dividends over the whole admitted range, dividends placed
next to the rounding boundaries of the quotient (where a merely faithful algorithm fails), zeros of both signs, infinities, NaN; phases next to the multiples
of 2 pi.  The device runs the same text (same builtins, -ffp-contract=off on both sides); the records of a loop built without it are compared byte for byte
on the GPU (tests/test_tracking_loop_gpu.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWO_PI = 2.0 * 3.1415926535898   # the reference's (GNSS ICD) pi, tracking_loop.hip


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("exact_division") / "libexact_division_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "gnss-sdr_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "exact_division_host.cc"), "-o", out], check=True)
    lib = C.CDLL(out)
    dp = C.POINTER(C.c_double)
    lib.gsh_test_div_by_constant.argtypes = [dp, C.c_int64, C.c_double, dp]
    lib.gsh_test_div_by_constant.restype = C.c_int64
    lib.gsh_test_fmod_by_constant.argtypes = [dp, C.c_int64, C.c_double, dp, C.POINTER(C.c_int64)]
    lib.gsh_test_fmod_by_constant.restype = C.c_int64
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _boundary_dividends(b, rng, n):
    """dividends whose quotient by b lies within a few ulp of a double (q b, rounded) or of the midpoint between two doubles ((q + ulp / 2) b): where a
    merely faithful quotient rounds the wrong way; each with its three neighbours on either side"""
    q = rng.uniform(1.0, 2.0, n) * 2.0 ** rng.integers(-40, 40, n)
    a = np.concatenate([q * b, q * b + (np.spacing(q) / 2) * b])
    out, up, down = [a], a, a
    for _ in range(3):
        up, down = np.nextafter(up, np.inf), np.nextafter(down, -np.inf)
        out += [up, down]
    return np.concatenate(out)


@pytest.mark.parametrize("b", [25e6, 4e6, 32e6, 50e6, 2.048e6, 1575.42e6, 1176.45e6, 1602e6 + 6 * 0.5625e6, TWO_PI, 3.0, 1.0, 7.0 / 3.0])
def test_division_by_a_constant_is_the_ieee_quotient(lib, b):
    rng = np.random.default_rng(int(b) % 9973)
    n = 1_000_000
    a = np.concatenate([
        rng.standard_normal(n) * 10.0 ** rng.uniform(-140, 140, n),     # the whole admitted range
        rng.standard_normal(n) * 1.0e6,                                 # Hz times chips/s scale
        rng.uniform(-1.0, 1.0, n) * 1.023e6,
        _boundary_dividends(b, rng, 200_000),
        -_boundary_dividends(b, rng, 50_000),
        np.arange(1, 200_001, dtype=np.float64) * b,                    # exact quotients
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, b, -b, b / 3, np.nextafter(b, 0.0), 1e-149, -1e149]),
    ]).astype(np.float64)
    a = np.ascontiguousarray(a)
    first = C.c_double(0.0)
    bad = lib.gsh_test_div_by_constant(_ptr(a), a.size, b, C.byref(first))
    assert bad == 0, f"{bad} of {a.size} quotients by {b!r} differ from a / b; first dividend {first.value!r}"


def test_fmod_by_two_pi_is_the_library_fmod(lib):
    rng = np.random.default_rng(11)
    n = 1_000_000
    k = rng.integers(0, 1000, n).astype(np.float64)
    near = k * TWO_PI
    x = np.concatenate([
        rng.uniform(-40.0, 40.0, n),                                    # the loop's range: a few turns
        rng.uniform(-1e5, 1e5, n) * TWO_PI,
        near, np.nextafter(near, np.inf), np.nextafter(near, -np.inf), -near,
        near + rng.uniform(-1e-9, 1e-9, n),
        rng.standard_normal(n).astype(np.float32).astype(np.float64) * 30.0,   # floats, as the loop's phase remainder is
        np.array([0.0, -0.0, TWO_PI, -TWO_PI, np.nextafter(TWO_PI, 0.0), np.nextafter(TWO_PI, 7.0), 5e-324, -5e-324, 1e-300, 6.2e6, 1e7, 1e300, np.inf, np.nan]),
    ]).astype(np.float64)
    x = np.ascontiguousarray(x)
    first = C.c_double(0.0)
    n_slow = C.c_int64(0)
    bad = lib.gsh_test_fmod_by_constant(_ptr(x), x.size, TWO_PI, C.byref(first), C.byref(n_slow))
    assert bad == 0, f"{bad} of {x.size} remainders differ from fmod(x, 2 pi); first argument {first.value!r}"
    assert 4 <= n_slow.value <= 10   # only the out-of-range tail of the list takes the library's path
