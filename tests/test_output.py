"""lt_get_output = the conversion loop of writeOutput (LTRANS.f90:1719-1723: x2lon, y2lat,
getStatus) and the columns the CSV / NetCDF writers take."""
import math

import numpy as np
import pytest

import cases
import oracle_lib
import ltrans_amd
from ltrans_amd.driver import LtransRun


def _oracle_kw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _run(case, kw, n_external=2, max_internal=None):   # 60 steps: past pediage and deadage
    e = cases.make_engine(case, ltrans_amd.Engine, kw)
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()
    run.run(n_external, max_internal=max_internal)
    return e


@pytest.mark.parametrize("spherical", [0, 1])
def test_oracle_output_formulas(spherical):
    case = cases.build_case("tiny_settle", SphericalProjection=spherical, lonmin=-77.0, latmin=36.5,
                            OpenOceanBoundary=1)
    e = _run(case, _oracle_kw())
    st = e.get_particles()
    out = e.get_output()
    c = case["cfg"]
    x, y = st["par"][:, 0], st["par"][:, 1]
    RCF = 180.0 / c.PI
    if spherical:
        lat = y * RCF / c.Earth_Radius + c.latmin
        lon = x * 180.0 / (c.Earth_Radius * c.PI * np.cos((y * 180.0 / (c.Earth_Radius * c.PI) + c.latmin) / RCF)) + c.lonmin
    else:
        lat = 2.0 * RCF * (np.arctan(np.exp(y / c.Earth_Radius)) - c.PI / 4.0)
        lon = x / c.Earth_Radius * RCF
    # (the Mercator latitude subtracts pi/4 from atan(exp(y/R)) ~ pi/4: ~1e3 ulp of cancellation)
    assert np.allclose(out["lon"], lon, rtol=1e-14, atol=0) and np.allclose(out["lat"], lat, rtol=1e-9, atol=1e-15)
    assert np.array_equal(out["z"], st["par"][:, 2]) and np.array_equal(out["age"], st["par"][:, 11])
    status = st["P_behave"].astype(np.float64)
    status[st["dead"] != 0] = -1
    status[st["settle"] != 0] = -2
    status[st["oob"] != 0] = -3
    assert np.array_equal(out["status"], status)
    assert (status == -2).any() and (status == -1).any()
    assert np.array_equal(out["hitBottom"], st["hitBottom"]) and out["hitBottom"].sum() > 0
    # printOutput clears the collision counters after writing them
    e.get_output(reset_hits=True)
    st2 = e.get_particles()
    assert st2["hitBottom"].sum() == 0 and st2["hitLand"].sum() == 0
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("spherical", [0, 1])
def test_hip_output_matches_oracle(spherical):
    case = cases.build_case("tiny_settle", SphericalProjection=spherical, lonmin=-77.0, latmin=36.5,
                            OpenOceanBoundary=1, SaltTempOn=1)
    eo = _run(case, _oracle_kw())
    eh = _run(case, {})
    oo, oh = eo.get_output(salt_temp=True), eh.get_output(salt_temp=True)
    for k in ("lon", "lat", "z", "age", "salt", "temp"):
        assert np.allclose(oh[k], oo[k], rtol=1e-9, atol=1e-13), k
    for k in ("status", "hitBottom", "hitLand"):
        assert np.array_equal(oh[k], oo[k]), k
    eh.get_output(reset_hits=True)
    assert eh.get_particles()["hitBottom"].sum() == 0
    eo.close()
    eh.close()
