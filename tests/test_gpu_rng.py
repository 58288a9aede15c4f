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


# ---- beyond 2^32 words (random_module.f90:166-197: the reference's stream has no position at all; the engine's is a
# 64-bit count -- configs[2] passes 2^32 words at step 35, a 3240-step run reaches 4e11)

def _ring(e, off, n):
    f = e.lib.lt_test_mt_ring
    f.restype = C.c_int
    out = np.zeros(n, dtype=np.uint32)
    assert f(e._h, C.c_uint64(off), C.c_int32(n), out.ctypes.data_as(C.c_void_p)) == 0
    return out


def _host_jump_words(st, offset, n):
    """n words `offset` draws after the next draw of state st, by the library's HOST polynomial jump
    (tests/test_mt_jump.py checks it against the sequential generator)"""
    lib = hip_test_lib.load()
    lib.lt_test_mt_stream.restype = C.c_int
    mt = np.array(st.mt, dtype=np.uint32).view(np.int32)
    out = np.zeros(n, dtype=np.uint32)
    assert lib.lt_test_mt_stream(mt.ctypes.data_as(C.c_void_p), C.c_int32(st.mti), C.c_uint64(offset), C.c_int32(n),
                                 out.ctypes.data_as(C.c_void_p)) == 0
    return out


def test_ring_past_two_to_the_32_against_the_sequential_generator(oracle):
    """5.0e9 words in: the oracle generates them one after the other (~20 s of host time)"""
    seed = 9
    e = hip_test_lib.HipTestEngine(ltrans_amd.LtConfig(numpar=4, us=8, ws=9, seed=seed))
    off = 5_000_000_321
    got = _ring(e, off, 3000)
    e.close()
    assert np.array_equal(got, _oracle_stream(oracle, seed, off, 3000))


@pytest.mark.parametrize("off", [(1 << 32) - 1500, 100_000_000_019, 400_000_000_777])
def test_ring_far_beyond_two_to_the_32_against_the_host_jump(oracle, off):
    """where a 3240-step run of configs[2] gets (4e11 words): the device's block generation and jumps against the
    host's jump from the seed state, also across the 2^32 boundary itself"""
    seed = 5489
    e = hip_test_lib.HipTestEngine(ltrans_amd.LtConfig(numpar=4, us=8, ws=9, seed=seed))
    got = _ring(e, off, 3000)
    e.close()
    st = oracle_lib.MT()
    oracle.lto_init_genrand(C.byref(st), C.c_uint32(seed))
    assert np.array_equal(got, _host_jump_words(st, off, 3000))


def _untemper(y):
    """inverse of MT19937's output tempering (random_module.f90:190-193)"""
    y = np.asarray(y, dtype=np.uint64)
    y = y ^ (y >> np.uint64(18))
    y = y ^ ((y << np.uint64(15)) & np.uint64(0xefc60000))
    t = y.copy()
    for _ in range(5):
        t = y ^ ((t << np.uint64(7)) & np.uint64(0x9d2c5680))
    y = t & np.uint64(0xffffffff)
    t = y.copy()
    for _ in range(3):
        t = y ^ (t >> np.uint64(11))
    return (t & np.uint64(0xffffffff)).astype(np.uint32)


def test_a_turbulence_step_whose_draws_straddle_two_to_the_32(oracle):
    """HTurb + VTurb with the stream's cursor moved to 2^32 - 2.5e6 (lt_test_rng_skip): 8192 particles draw 1.0e6
    words per step, so the third step's window of the stream contains word 2^32.  The oracle starts from the state
    at that position (host jump + untempering of the 624 words before it).  Bit-identical particles, the same
    stream position afterwards."""
    import cases
    from ltrans_amd.driver import LtransRun
    case = cases.build_case("mid_turb", numpar=8192)
    seed = case["cfg"].seed
    skip = (1 << 32) - 2_500_000 - 624           # cursor after init_genrand = 624
    st = oracle_lib.MT()
    oracle.lto_init_genrand(C.byref(st), C.c_uint32(seed))
    assert st.mti == 624
    state = _untemper(_host_jump_words(st, skip - 624, 624))   # the 624 words before the first one to be drawn
    chk = oracle_lib.MT()
    for i in range(624):
        chk.mt[i] = int(state[i])
    chk.mti = 624
    want_first = _host_jump_words(st, skip, 4)
    assert [oracle.lto_genrand_int32(C.byref(chk)) for _ in range(4)] == [int(w) for w in want_first]
    outs = []
    for which in ("hip", "oracle"):
        if which == "hip":
            e = cases.make_engine(case, hip_test_lib.HipTestEngine, {})
            f = e.lib.lt_test_rng_skip
            f.restype = C.c_int
            assert f(e._h, C.c_uint64(skip)) == 0
        else:
            e = cases.make_engine(case, ltrans_amd.Engine, dict(lib=oracle, prefix="lto_"))
            e.set_rng(state, 624)
        run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
        run.init_hydro()
        run.run(1, max_internal=5)
        out = e.get_particles()
        out["draws"] = e.stats()["rng_draws"]
        mt, mti = e.get_rng()
        nxt = oracle_lib.MT()
        for i in range(624):
            nxt.mt[i] = int(mt[i])
        nxt.mti = int(mti)
        out["next"] = [oracle.lto_genrand_int32(C.byref(nxt)) for _ in range(8)]
        e.close()
        outs.append(out)
    hip, ora = outs
    assert hip["draws"] == ora["draws"] == 8192 * 124
    assert np.array_equal(hip["par"], ora["par"])
    for k in ("P_r_element", "err_code", "hitBottom", "hitLand"):
        assert np.array_equal(hip[k], ora[k]), k
    assert hip["next"] == ora["next"]
