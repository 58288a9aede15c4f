"""The materialised MT19937 stream on the device (block generation + (P-1)*S
jump-ahead per workgroup) against the oracle's sequential generator: bit-exact,
including reads that straddle block boundaries and blocks reached only through
jumps."""
import ctypes as C

import numpy as np
import pytest

import hip_test_lib
import ltrans_amd
import oracle_lib

pytestmark = pytest.mark.gpu

S = 1 << 20
P = 128


def _oracle_stream(oracle, seed, skip, n):
    st = oracle_lib.MT()
    oracle.lto_init_genrand(C.byref(st), C.c_uint32(seed))
    out = np.zeros(n, dtype=np.uint32)
    oracle.lto_genrand_fill(C.byref(st), C.c_uint64(skip), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


@pytest.mark.parametrize("seed", [9, 5489])
def test_ring_matches_sequential_stream(oracle, seed):
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9, seed=seed)
    e = hip_test_lib.HipTestEngine(cfg)
    f = e.lib.lt_test_mt_ring
    f.restype = C.c_int
    # offsets relative to the cursor (= first draw after init_genrand)
    offsets = [0, 100, S - 624 - 50, S - 10, 3 * S + 17, (P - 1) * S - 5, P * S - 300,
               (P + 2) * S + 12345, (2 * P + 1) * S - 100]
    for off in offsets:
        out = np.zeros(1000, dtype=np.uint32)
        rc = f(e._h, C.c_uint64(off), C.c_int32(out.size), out.ctypes.data_as(C.c_void_p))
        assert rc == 0, e.lib.lt_last_error
        assert np.array_equal(out, _oracle_stream(oracle, seed, off, out.size)), off
    mt, mti = e.get_rng()
    e.close()


def test_set_rng_mid_batch(oracle):
    """a state handed over with mti < 624 (part of the array still unconsumed)."""
    st = oracle_lib.MT()
    oracle.lto_init_genrand(C.byref(st), C.c_uint32(77))
    burn = np.zeros(1000, dtype=np.uint32)
    oracle.lto_genrand_fill(C.byref(st), C.c_uint64(0), C.c_int(1000), burn.ctypes.data_as(C.c_void_p))
    assert 0 < st.mti < 624
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9)
    e = hip_test_lib.HipTestEngine(cfg)
    e.set_rng(np.array(st.mt, dtype=np.uint32), st.mti)
    f = e.lib.lt_test_mt_ring
    f.restype = C.c_int
    out = np.zeros(3000, dtype=np.uint32)
    assert f(e._h, C.c_uint64(0), C.c_int32(out.size), out.ctypes.data_as(C.c_void_p)) == 0
    exp = np.zeros(3000, dtype=np.uint32)
    oracle.lto_genrand_fill(C.byref(st), C.c_uint64(0), C.c_int(3000), exp.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, exp)
    e.close()


def test_ring_far_into_the_stream(oracle):
    """3e9 words in (past 2^31: 64-bit stream positions, ~2900 blocks, every generator
    state jumped ~22 times), as a long turbulence run gets after a few dozen steps"""
    seed = 9
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9, seed=seed)
    e = hip_test_lib.HipTestEngine(cfg)
    f = e.lib.lt_test_mt_ring
    f.restype = C.c_int
    off = 3_000_000_123
    out = np.zeros(2000, dtype=np.uint32)
    assert f(e._h, C.c_uint64(off), C.c_int32(out.size), out.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(out, _oracle_stream(oracle, seed, off, out.size))
    e.close()
