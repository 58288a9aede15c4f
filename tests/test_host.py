"""Host logic around the engine: LTRANS.data reader, time-step loops, the
init-time geometry stand-ins, sharding helpers; and whole-step sanity of the CPU
oracle on the synthetic inputs (the checker must itself behave)."""
import os

import math

import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib
from ltrans_amd import roms_synth as rs
from ltrans_amd import sharding
from ltrans_amd.driver import LtransRun

HERE = os.path.dirname(os.path.abspath(__file__))


def test_namelist_reader():
    g = ltrans_amd.read_namelists(os.path.join(HERE, "sample_LTRANS.data"))
    assert g["numparticles"]["numpar"] == 1234
    assert g["romsgrid"]["ncgridfile"] == "./input/some_grid.nc"
    c = ltrans_amd.config_from_namelists(g)
    assert (c.numpar, c.dt, c.idt, c.us, c.ws, c.Vtransform) == (1234, 1800, 60, 12, 13, 2)
    assert c.hc == np.float32(0.5) and c.z0 == 0.001
    assert (c.HTurbOn, c.VTurbOn, c.ConstantHTurb) == (0, 0, 2.5)
    assert (c.Behavior, c.OpenOceanBoundary, c.mortality) == (6, 1, 0)
    assert (c.swimstart, c.sink, c.Swimdepth) == (3600.0, -0.0005, 2.0)
    assert (c.settlementon, c.holesExist, c.seed, c.ErrorFlag, c.TrackCollisions) == (1, 0, 42, 2, 1)


def test_namelist_reader_on_reference_file():
    ref = "/root/reference/Model/LTRANS.data"
    if not os.path.exists(ref):
        pytest.skip("reference not present")
    c = ltrans_amd.config_from_namelists(ltrans_amd.read_namelists(ref))
    assert (c.numpar, c.us, c.ws, c.dt, c.idt, c.seed, c.Behavior) == (608, 20, 21, 3600, 120, 9, 4)
    assert c.hc == np.float32(0.2) and c.PI == 3.14159265358979


def test_time_arrays_follow_the_reference_loops():
    run = LtransRun(None, 3600, 120, None)
    ex = run.external_times(3)                     # LTRANS.f90:568-571
    assert list(ex) == [3600.0, 7200.0, 10800.0]
    assert list(run.internal_times(ex, 1)) == [7200.0 - 120.0, 7200.0, 7320.0]   # :588-590
    assert list(run.internal_times(ex, 30)) == [7200.0 + 28 * 120, 7200.0 + 29 * 120, 10800.0]


def test_geometry_conventions():
    g = rs.make_grid(20, 14, 6, islands=[(8, 10, 5, 7)])
    geo = rs.build_geometry(g)
    vi = g["vi"]
    RE, adj = geo["RE"], geo["r_Adjacent"]
    assert np.array_equal(RE[1], RE[0] + 1) and np.array_equal(RE[2], RE[0] + 1 + vi)
    assert np.array_equal(RE[3], RE[0] + vi)                      # corner order :420-423
    assert np.array_equal(adj[:, 0], np.arange(1, RE.shape[1] + 1))   # self first :592
    for row in adj[:50]:
        nz = row[1:][row[1:] > 0]
        assert np.all(np.diff(nz) > 0) and np.all(row[1 + nz.size:] == 0)   # ascending, zero tail
        for j in nz:                                              # neighbours share a node
            assert set(RE[:, row[0] - 1]) & set(RE[:, j - 1])
    assert geo["rho_elements"] < (vi - 1) * (g["uj"] - 1)         # the island's dry elements dropped
    b = rs.create_bounds(g)
    assert b["numislands"] == 1 and b["nbounds"] == b["maxbound"] + b["maxisland"] - 2
    assert (b["bx"][0], b["by"][0]) == (b["bx"][-1], b["by"][-1])   # closed polygons
    assert (b["hx"][0], b["hy"][0]) == (b["hx"][-1], b["hy"][-1])
    # every segment joins consecutive polygon points (boundary_module.f90:1161-1201)
    assert np.array_equal(b["bnd_x"][0, :b["maxbound"] - 1], b["bx"][:-1])
    assert np.array_equal(b["bnd_x"][1, b["maxbound"] - 1:], b["hx"][1:])


def test_particles_start_inside_their_elements(oracle):
    g = rs.make_grid(40, 30, 6, islands=[(15, 18, 10, 13)])
    geo = rs.build_geometry(g)
    b = rs.create_bounds(g)
    P = rs.make_particles(g, geo, b, 500, near_land=0.3)
    O = oracle_lib.Leaf(oracle, "o")
    ex, ey = geo["rx"][geo["RE"] - 1], geo["ry"][geo["RE"] - 1]
    for n in range(0, 500, 7):
        e = int(P["P_r_element"][n])
        assert O.gridcell(ex, ey, P["par"][n, 0], P["par"][n, 1], e) == (e, 1)
    bl = np.c_[b["bx"], b["by"]]
    assert all(O.inpoly(P["par"][n, 0], P["par"][n, 1], bl) == 1 for n in range(0, 500, 7))


def test_shard_ranges_partition_the_particles():
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (608, 2)):
        r = [sharding.shard_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _oracle_engine(case):
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def test_oracle_step_is_sane():
    case = cases.build_case("tiny_adv")
    out = cases.run_case(case, ltrans_amd.Engine, _oracle_engine(case), n_external=2)
    par = out["par"]
    g = case["g"]
    assert out["stats"]["active"] == case["cfg"].numpar and out["err_code"].max() == 0
    assert np.all(par[:, 11] == 60 * 120.0)                       # age advanced by idt each step
    assert np.all(par[:, 0] > 0) and np.all(par[:, 0] < (g["vi"] - 1) * g["dx"])
    assert np.all(par[:, 2] < 1.0) and np.all(par[:, 2] > -g["h"].max())
    moved = np.hypot(par[:, 0] - case["parts"]["par"][:, 0], par[:, 1] - case["parts"]["par"][:, 1])
    assert moved.max() > 500.0                                    # currents of O(0.3 m/s) for 2 h


def test_oracle_unreleased_particles_hold_position():
    case = cases.build_case("tiny_adv")
    case["parts"]["par"][::2, 10] = 1e9                           # DOB in the future
    out = cases.run_case(case, ltrans_amd.Engine, _oracle_engine(case), n_external=1, max_internal=5)
    assert np.array_equal(out["par"][::2, :3], case["parts"]["par"][::2, :3])
    assert np.all(out["par"][::2, 11] == 0.0) and np.all(out["par"][1::2, 11] == 600.0)
    assert out["stats"]["active"] == case["cfg"].numpar // 2


def test_exp_table_header_is_the_generated_one():
    """lt_exp_table.h (the 2^(k/128) table of the libm-exact device exp) is what
    gen_exp_table.py computes from the definition, and reproduces libm's exp on the host when the
    published evaluation order is followed with it (pure-Python fma is not available, so the check
    here is the table itself plus two anchor identities)."""
    import importlib.util
    import struct
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ltransv.2b_amd", "csrc")
    spec = importlib.util.spec_from_file_location("gen_exp_table", os.path.join(csrc, "gen_exp_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(csrc, "lt_exp_table.h")) as f:
        assert f.read() == gen.render()
    tab = gen.table()
    assert len(tab) == 128 and tab[0] == (0, 0x3FF0000000000000)
    for k in (1, 37, 64, 127):
        tail, sbits = tab[k]
        h = struct.unpack("<d", struct.pack("<Q", sbits + (k << 45)))[0]
        assert h == 2.0 ** (k / 128.0) or abs(h - 2.0 ** (k / 128.0)) <= 2.3e-16 * h
        assert abs(struct.unpack("<d", struct.pack("<Q", tail))[0]) < 2.0 ** -53


def test_division_by_a_shared_constant_is_exact_in_rational_arithmetic():
    """div_by (csrc/lt_math.h): a / b for a divisor every particle shares as q = RN(a*h + RN(a*l)) with the host's
    double-double h + l = 1/b (lt_div_const, no GPU needed).  Checked here in exact rational arithmetic -- Python's
    Fraction -> float conversion rounds correctly -- against RN(a/b) on numerators built to put the quotient next to
    the midpoint of two doubles (where a wrong last bit would show), on random ones, and on the sign of zero; and the
    divisors lt_div_const must refuse."""
    import ctypes as C
    import random
    from fractions import Fraction
    import ltrans_amd
    lib = ltrans_amd.load_hip_library()
    lib.lt_div_const.restype = C.c_int
    lib.lt_div_const.argtypes = [C.c_double, C.POINTER(C.c_double)]

    def consts(b):
        hl = (C.c_double * 2)()
        return lib.lt_div_const(b, hl), hl[0], hl[1]

    def algo(a, h, l):
        u1 = float(Fraction(a) * Fraction(l))            # RN(a * l)
        return float(Fraction(a) * Fraction(h) + Fraction(u1))   # RN(a * h + u1): the FMA

    rnd = random.Random(5)
    for b in (124.0, 204.0, 6.0, -3600.0, -7200.0, 3600.0, 120.0, 36.0, 84.0, 44.0, -86400.0, 3.0, 1.0, 0.5, -2.0,
              float((1 << 30) - 1), 900.0 * 2.0 ** -40):
        ok, h, l = consts(b)
        assert ok == 1, b
        assert (h < 0) == (b < 0) and (math.copysign(1.0, l) < 0) == (b < 0)
        assert abs(Fraction(1) / Fraction(b) - Fraction(h) - Fraction(l)) <= abs(Fraction(1) / Fraction(b)) * Fraction(1, 2 ** 104)
        nums = [rnd.uniform(-1e6, 1e6) for _ in range(300)] + [rnd.uniform(-1, 1) * 10.0 ** rnd.randint(-200, 200) for _ in range(300)]
        # quotients next to midpoints: a = RN(b * (an odd multiple of half an ulp)), and its neighbours
        for _ in range(400):
            m = rnd.getrandbits(52) | (1 << 52)
            e = rnd.randint(-60, 60)
            mid = Fraction(2 * m + 1, 2) * Fraction(2) ** e
            a0 = float(mid * Fraction(b))
            nums += [a0, math.nextafter(a0, math.inf), math.nextafter(a0, -math.inf)]
        for a in nums:
            want = float(Fraction(a) / Fraction(b))
            got = algo(a, h, l)
            assert got == want and math.copysign(1.0, got) == math.copysign(1.0, want), (a, b, got, want)
        # zero numerators: the sign of the quotient (both parts of the constant carry b's sign)
        for a in (0.0, -0.0):
            want = a / b
            got = a * h + a * l
            assert got == 0.0 and math.copysign(1.0, got) == math.copysign(1.0, want), (a, b)
    for b in (0.0, math.inf, math.nan, 1e200, 1e-200, math.pi, float((1 << 31) + 1), 1.0 / 3.0):
        assert consts(b)[0] == 0, b
