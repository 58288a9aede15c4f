"""The prepared-divisor division used in the level-depth and YPC1 arithmetic
(ltransv.2b_amd/csrc/lt_math.h: prep_div / div_prep) must be bit-identical to
the compiler's `a / b` on the operand ranges it is used for."""
import ctypes as C

import numpy as np
import pytest

import ltrans_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip_lib():
    lib = ltrans_amd.load_hip_library()
    lib.lt_test_div.restype = C.c_int
    return lib


def _run(lib, a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    qf = np.empty_like(a)
    qr = np.empty_like(a)
    rc = lib.lt_test_div(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                         qf.ctypes.data_as(C.c_void_p), qr.ctypes.data_as(C.c_void_p), C.c_int32(a.size))
    assert rc == 0
    return qf, qr


def test_prepared_division_is_exact(hip_lib):
    rng = np.random.default_rng(7)
    n = 4_000_000
    # divisors: depths / level spacings, 1e-9 .. 1e5 m, both signs; random mantissas
    b = np.exp(rng.uniform(np.log(1e-9), np.log(1e5), n)) * rng.choice([-1.0, 1.0], n)
    # numerators: differences and products of fp32 field values, 1e-60 .. 1e8, and exact zeros
    a = np.exp(rng.uniform(np.log(1e-60), np.log(1e8), n)) * rng.choice([-1.0, 1.0], n)
    a[::97] = 0.0
    a[1::97] = -0.0
    qf, qr = _run(hip_lib, a, b)
    assert np.array_equal(qf.view(np.uint64), qr.view(np.uint64))
    # the host's IEEE division agrees with both
    assert np.array_equal(qr, a / b)


def test_prepared_division_near_one_and_fp32_values(hip_lib):
    rng = np.random.default_rng(8)
    n = 2_000_000
    b = rng.uniform(0.5, 2.0, n)
    a = (rng.standard_normal(n).astype(np.float32)).astype(np.float64) * rng.uniform(0.0, 1.0, n)
    qf, qr = _run(hip_lib, a, b)
    assert np.array_equal(qf.view(np.uint64), qr.view(np.uint64))
    # quotients that are exactly representable, and divisors with long runs of ones
    b2 = np.nextafter(np.ldexp(1.0, rng.integers(-20, 20, n)), 0.0)
    a2 = b2 * rng.integers(1, 1 << 20, n)
    qf, qr = _run(hip_lib, a2, b2)
    assert np.array_equal(qf.view(np.uint64), qr.view(np.uint64))
