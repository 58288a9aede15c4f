"""MT19937 jump-ahead (characteristic polynomial by Berlekamp-Massey, x^J mod phi,
window evaluation) of the product library, on the HOST, against the oracle's
sequential generator: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import hip_test_lib
import ltrans_amd
import oracle_lib


def _seed_state(oracle, seed, burn=0):
    st = oracle_lib.MT()
    oracle.lto_init_genrand(C.byref(st), C.c_uint32(seed))
    for _ in range(burn):
        oracle.lto_genrand_int32(C.byref(st))
    return st


def _oracle_words(oracle, st, skip, n):
    s2 = oracle_lib.MT()
    C.memmove(C.byref(s2), C.byref(st), C.sizeof(st))
    for _ in range(skip):
        oracle.lto_genrand_int32(C.byref(s2))
    return np.array([oracle.lto_genrand_int32(C.byref(s2)) for _ in range(n)], dtype=np.uint32)


@pytest.mark.parametrize("seed,burn,offset", [(9, 0, 0), (9, 0, 5000), (5489, 100, 123457),
                                               (4357, 700, 1 << 20), (9, 3, 624)])
def test_host_jump_matches_sequential(oracle, seed, burn, offset):
    lib = hip_test_lib.load()
    st = _seed_state(oracle, seed, burn)
    mt = np.array(st.mt, dtype=np.uint32).view(np.int32)
    out = np.zeros(700, dtype=np.uint32)
    rc = lib.lt_test_mt_stream(mt.ctypes.data_as(C.c_void_p), C.c_int32(st.mti), C.c_uint64(offset),
                               C.c_int32(out.size), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.array_equal(out, _oracle_words(oracle, st, offset, out.size))


@pytest.mark.parametrize("a,b", [(1, 1), (624, 19937), ((1 << 20) * 255, 1 << 20), ((1 << 20) * 1279, (1 << 20) * 256),
                                 (123456789, 987654321)])
def test_product_of_jumps_is_the_jump_over_the_sum(a, b):
    """poly_mulmod: the engine forms the jump polynomials of a shard (j*P and j*P-1 blocks) as products of kept
    powers of x^(P*S) with x^((P-1)*S)"""
    lib = hip_test_lib.load()
    lib.lt_test_mt_mulmod.restype = C.c_int
    assert lib.lt_test_mt_mulmod(C.c_uint64(a), C.c_uint64(b)) == 1
