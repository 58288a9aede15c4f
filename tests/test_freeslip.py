"""FreeSlip (hydrodynamic_module.f90:1934-1994 in getInterp, :2329-2520 in interp): values on the land
nodes of an element are replaced by values of its water nodes before the interpolation.  CPU side:
the oracle's restatement, including the two slips of the reference's u / v branches."""
import ctypes as C

import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib


@pytest.fixture(scope="module")
def okw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _py_sub(kind, m, d, v):
    """independent restatement of the substitution (m, d: masks of the 4 nodes, 1 = water)"""
    v = list(v)
    s = sum(m)
    if s >= 4:
        return v
    o = list(v)
    if s == (1 if kind == 1 else 3):
        for k, (a, b) in enumerate(((1, 3), (0, 2), (1, 3), (0, 2))):
            if m[k] == 0:
                v[k] = 0.5 * (o[a] + o[b])
                break
    elif s == 2:
        pairs = (((0, 1), ((0, 3), (1, 2))), ((1, 2), ((1, 0), (2, 3))), ((2, 3), ((2, 1), (3, 0))),
                 ((3, 0), ((3, 2), (0, 1))))
        for (a, b), copies in pairs:
            if m[a] == 0 and m[b] == 0:
                for dst, src in copies:
                    v[dst] = o[src]
                return v
        a = d if kind == 2 else m
        if a[0] == 0 and a[2] == 0:
            v[0], v[2] = o[3], o[1]
        elif a[3] == 0 and a[1] == 0:
            v[3], v[1] = o[0], o[2]
    elif s == 1 and kind != 1:
        k = m.index(1)
        v = [o[k]] * 4
    return v


def _all_cases():
    rows = []
    for kind in (0, 1, 2):
        for m in range(16):
            for d in range(16):
                if kind != 2 and d != 15:
                    continue
                rows.append((kind, m, d))
    return rows


def test_substitution_table(okw):
    """every mask combination of the three variants: the oracle against the independent restatement"""
    lib = okw["lib"]
    lib.lto_freeslip_sub.restype = None
    v0 = [9.0, 2.0, 7.0, 4.0]
    for kind, m, d in _all_cases():
        mm = [(m >> k) & 1 for k in range(4)]
        dd = [(d >> k) & 1 for k in range(4)]
        v = (C.c_double * 4)(*v0)
        lib.lto_freeslip_sub(C.c_int(kind), (C.c_int32 * 4)(*mm), (C.c_int32 * 4)(*dd), v)
        assert list(v) == _py_sub(kind, mm, dd, v0), (kind, mm, dd)
    # the reference's slips, spelled out
    assert _py_sub(0, [0, 1, 1, 1], None, [9.0, 2.0, 3.0, 4.0]) == [3.0, 2.0, 3.0, 4.0]
    assert _py_sub(1, [0, 1, 1, 1], None, [9.0, 2.0, 3.0, 4.0]) == [9.0, 2.0, 3.0, 4.0]      # u grid: left alone
    assert _py_sub(1, [0, 0, 0, 1], None, [9.0, 8.0, 7.0, 4.0]) == [6.0, 8.0, 7.0, 4.0]      # u grid: sum == 1
    assert _py_sub(0, [0, 0, 0, 1], None, [9.0, 8.0, 7.0, 4.0]) == [4.0, 4.0, 4.0, 4.0]
    assert _py_sub(2, [0, 1, 0, 1], [1, 1, 1, 1], [9.0, 2.0, 7.0, 4.0]) == [9.0, 2.0, 7.0, 4.0]  # v grid reads d
    assert _py_sub(2, [0, 1, 0, 1], [1, 0, 1, 0], [9.0, 2.0, 7.0, 4.0]) == [9.0, 7.0, 7.0, 9.0]


def test_freeslip_acts_only_next_to_land(okw, name="tiny_fs_adv"):
    """with FreeSlip the particles that stay in all-water elements move exactly as without it, the others do not"""
    on = cases.run_case(cases.build_case(name), ltrans_amd.Engine, okw, n_external=1, max_internal=4)
    off = cases.run_case(cases.build_case(name, FreeSlip=0), ltrans_amd.Engine, okw, n_external=1, max_internal=4)
    same = np.all(on["par"][:, :3] == off["par"][:, :3], axis=1)
    assert 0 < np.count_nonzero(~same) < same.size
    case = cases.build_case(name)
    geo = case["geo"]
    wet = lambda E, mask, el: np.all(mask[E[:, el - 1] - 1] == 1, axis=0)  # noqa: E731
    inland = (wet(geo["RE"], geo["rho_mask"], off["P_r_element"]) & wet(geo["UE"], geo["u_mask"], off["P_u_element"])
              & wet(geo["VE"], geo["v_mask"], off["P_v_element"])
              & wet(geo["RE"], geo["rho_mask"], case["parts"]["P_r_element"])
              & wet(geo["UE"], geo["u_mask"], case["parts"]["P_u_element"])
              & wet(geo["VE"], geo["v_mask"], case["parts"]["P_v_element"]))
    # a particle that started and ended in all-water elements may still have crossed a coastal one;
    # the converse must hold: whoever differs touched an element with a land node
    assert np.count_nonzero(~same & inland) <= np.count_nonzero(~same) // 2
    assert np.count_nonzero(~same & ~inland) > 0


def test_freeslip_needs_masks(okw):
    case = cases.build_case("tiny_fs_adv")
    geo = dict(case["geo"])
    geo["rho_mask"] = None
    e = oracle_lib.engine(case["cfg"], **okw)
    with pytest.raises(ltrans_amd.LtError):
        e.set_grid(geo)


def test_freeslip_reaches_salinity_and_turbulence(okw):
    """WCTS_ITPI("salt"/"temp") and VTurb's Kh column go through the same substitution"""
    on = cases.run_case(cases.build_case("tiny_fs_salt"), ltrans_amd.Engine, okw, n_external=1, max_internal=2)
    off = cases.run_case(cases.build_case("tiny_fs_salt", FreeSlip=0), ltrans_amd.Engine, okw, n_external=1, max_internal=2)
    assert np.count_nonzero(on["P_Salt"] != off["P_Salt"]) > 0
    on = cases.run_case(cases.build_case("tiny_fs_turb"), ltrans_amd.Engine, okw, n_external=1, max_internal=2)
    off = cases.run_case(cases.build_case("tiny_fs_turb", FreeSlip=0), ltrans_amd.Engine, okw, n_external=1, max_internal=2)
    assert np.count_nonzero(on["par"][:, 2] != off["par"][:, 2]) > 0
