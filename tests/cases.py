"""Shared parity-test cases: synthetic ROMS inputs + the driver loop, runnable
against any implementation of the C ABI (HIP engine or CPU oracle)."""
import numpy as np

import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun

CASES = {
    # name: grid (vi,uj,us), islands, numpar, cfg overrides, near_land
    "tiny_adv": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                     cfg=dict(Behavior=0)),
    "tiny_sink": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=6, sink=-0.003, swimstart=0.0)),
    "small_adv": dict(grid=(100, 100, 20), islands=[(40, 44, 50, 53), (70, 71, 20, 30)],
                      numpar=608, near_land=0.2, cfg=dict(Behavior=0)),
    "mid_adv": dict(grid=(128, 64, 30), islands=[], numpar=20000, near_land=0.1,
                    cfg=dict(Behavior=0)),
    "mid_turb": dict(grid=(128, 64, 30), islands=[], numpar=20000, near_land=0.1,
                     cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=1)),
    "tiny_vt2": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                     cfg=dict(Behavior=0, Vtransform=2, hc=5.0)),
    "tiny_vt3": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                     cfg=dict(Behavior=0, Vtransform=3)),
    # random-walk turbulence: Mersenne-Twister stream consumed in particle order
    "tiny_turb": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0)),
    "tiny_hturb": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.5,
                       cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=0, ConstantHTurb=20.0)),
    "tiny_vturb": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                       cfg=dict(Behavior=6, sink=-0.001, HTurbOn=0, VTurbOn=1)),
    "tiny_turb_vt2": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                          cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0, Vtransform=2, hc=5.0)),
    "tiny_turb_vt3": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                          cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0, Vtransform=3)),
    "small_turb": dict(grid=(100, 100, 20), islands=[(40, 44, 50, 53), (70, 71, 20, 30)],
                       numpar=608, near_land=0.2, cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=1)),
    # biological behaviours 1-5 and 7 (behavior_module.f90:181-551): genrand_real1 draws in
    # particle order (a data-dependent number for 4 and 5), salinity cue through WCTS_ITPI
    "tiny_beh1": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=1, swimstart=0.0, swimslow=0.002, swimfast=0.004, pediage=7200.0)),
    "tiny_beh2": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=2, swimstart=0.0, swimslow=0.002, swimfast=0.004, pediage=7200.0,
                               HTurbOn=1, ConstantHTurb=1.0)),
    "tiny_beh3": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=3, swimstart=0.0, swimslow=0.002, swimfast=0.004, pediage=7200.0,
                               twistart=0.5, twiend=3.0, daylength=2.5, Kd=0.2)),
    "tiny_beh4": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=4, swimstart=0.0, swimslow=0.003, swimfast=0.006, pediage=9000.0,
                               deadage=20000.0, Sgradient=0.3, VTurbOn=1)),
    "tiny_beh5": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=5, swimstart=0.0, swimslow=0.003, swimfast=0.006, pediage=9000.0,
                               deadage=20000.0, Sgradient=0.3)),
    # C. virginica / C. ariakensis late stages: age-dependent switch, the 2-h bottom timer,
    # the change to pediveliger behaviour (P_behave 4/5 -> 2) at pediage
    "tiny_beh4_old": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                          age0=(100000.0, 800000.0),
                          cfg=dict(Behavior=4, swimstart=0.0, swimslow=0.003, swimfast=0.006, pediage=760000.0,
                                   deadage=900000.0, Sgradient=0.3)),
    "tiny_beh5_old": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                          age0=(100000.0, 420000.0),
                          cfg=dict(Behavior=5, swimstart=0.0, swimslow=0.003, swimfast=0.006, pediage=400000.0,
                                   deadage=900000.0, Sgradient=0.3, HTurbOn=1)),
    "tiny_beh7": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=7, Hswimspeed=0.3, Swimdepth=2.0)),
    "tiny_salt": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.3,
                      cfg=dict(Behavior=0, SaltTempOn=1)),
    # uniform drift into the west/south coast and the island: many reflections
    "tiny_reflect": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=768, near_land=0.6,
                         drift=(-0.35, -0.25), cfg=dict(Behavior=0)),
    # free slip (hydrodynamic_module.f90:1934-1994, 2329-2520): many particles in elements with land nodes
    "tiny_fs_adv": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=768, near_land=0.6,
                        drift=(-0.2, -0.15), cfg=dict(Behavior=0, FreeSlip=1)),
    "tiny_fs_turb": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.6,
                         cfg=dict(Behavior=0, FreeSlip=1, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0)),
    "tiny_fs_salt": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=512, near_land=0.6,
                         cfg=dict(Behavior=5, FreeSlip=1, SaltTempOn=1, swimstart=0.0, swimslow=0.002, swimfast=0.004,
                                  Sgradient=0.3)),
    # horizontal random walk with steps of more than a cell: particles jump over elements (errors 27-29 of
    # the closing setEle, LTRANS.f90:1317-1364), leave the domain (25), land on the island (26)
    "tiny_jump": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=768, near_land=0.2,
                      cfg=dict(Behavior=0, HTurbOn=1, VTurbOn=0, ConstantHTurb=6000.0)),
    # open eastern boundary, drift towards it: particles leave through open ocean
    "tiny_open": dict(grid=(24, 16, 8), islands=[], numpar=512, near_land=0.0, open_east=True,
                      drift=(0.45, 0.05), cfg=dict(Behavior=0, OpenOceanBoundary=1)),
    # sinking + settlement on habitat polygons (one with a hole), mortality on
    "tiny_settle": dict(grid=(24, 16, 8), islands=[(10, 12, 6, 8)], numpar=768, near_land=0.2,
                        polygons=[(6000.0, 5000.0, 2500.0, 12), (16000.0, 9000.0, 3000.0, 9),
                                  (18000.0, 4000.0, 1500.0, 7)],
                        holes=[(0, 6200.0, 5200.0, 900.0, 8)],
                        cfg=dict(Behavior=6, sink=-0.002, swimstart=0.0, settlementon=1,
                                 holesExist=1, pediage=1800.0, mortality=1, deadage=5400.0)),
}


def build_case(name, numpar=None, slab_storage=None, **cfg_extra):
    spec = CASES[name]
    vi, uj, us = spec["grid"]
    g = rs.make_grid(vi, uj, us, islands=spec["islands"], open_east=spec.get("open_east", False))
    geo = rs.build_geometry(g)
    bounds = rs.create_bounds(g)
    n = numpar or spec["numpar"]
    kw = dict(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, TrackCollisions=1)
    kw.update(spec["cfg"])
    kw.update(cfg_extra)
    if slab_storage is not None:
        kw["slab_storage"] = slab_storage
    cfg = ltrans_amd.LtConfig(**kw)
    parts = rs.make_particles(g, geo, bounds, n, seed=12345, near_land=spec["near_land"])
    parts.update(rs.behaviour_state(cfg, n))
    if "age0" in spec:       # particles that are already old: late-stage behaviour branches
        lo, hi = spec["age0"]
        parts["par"][:, 11] = np.linspace(lo, hi, n)          # par(:,pAge)
        parts["par"][:, 10] = -parts["par"][:, 11] - 1.0       # par(:,pDOB): released before t = 0
    habitat = None
    if cfg.settlementon:
        habitat = rs.make_habitat(g, geo, spec["polygons"], spec.get("holes", ()))
    recs = {}

    def record(r):
        if r not in recs:
            recs[r] = rs.make_record(g, geo, r, with_kh=True, drift=spec.get("drift", (0.0, 0.0)))
        return recs[r]

    roms = {}

    def rom(r, dtype=np.float64):
        """the same record as the file delivers it (unmasked), for the slab pipeline"""
        if (r, dtype) not in roms:
            roms[(r, dtype)] = rs.make_record(g, geo, r, with_kh=True, drift=spec.get("drift", (0.0, 0.0)),
                                              raw=dtype)
        return roms[(r, dtype)]

    return dict(name=name, g=g, geo=geo, bounds=bounds, cfg=cfg, parts=parts, record=record,
                habitat=habitat, rom=rom, dims=(g["vi"], g["uj"], g["ui"], g["vj"]))


def make_engine(case, engine_cls, engine_kw):
    import oracle_lib
    e = oracle_lib.engine(case["cfg"], **engine_kw) if engine_cls is ltrans_amd.Engine else engine_cls(case["cfg"], **engine_kw)
    e.set_grid(case["geo"])
    e.set_bounds(case["bounds"])
    if case["habitat"] is not None:
        e.set_habitat(case["habitat"])
    if case.get("habitat_raw") is not None:     # the lists built by the engine (createPolySpecs)
        e.create_polyspecs(case["habitat_raw"])
    e.set_particles(**case["parts"])
    return e


def run_case(case, engine_cls, engine_kw, n_external=2, max_internal=None, trace=None,
             pipeline=None, window_fn=None):
    """pipeline = np.float64 / np.float32: updateHydro through slab_prefetch / slab_commit
    with host arrays of that type instead of set_slab."""
    e = make_engine(case, engine_cls, engine_kw)
    if pipeline is None:
        run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    else:
        run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"],
                        rom_fn=lambda r: case["rom"](r, pipeline), dims=case["dims"], window_fn=window_fn)
    run.init_hydro()

    def on_step(p, it, ex, ix):
        if trace is not None:
            trace.append(e.get_particles(fields=("par", "err_code")))

    run.run(n_external, max_internal=max_internal, on_step=on_step if trace is not None else None)
    out = e.get_particles()
    out["stats"] = e.stats()
    if hasattr(e, "counters") and getattr(e, "prefix", "") == "lt_":
        out["counters"] = e.counters()
    e.close()
    return out


def rel_error(par_a, par_b):
    """max over particles and x,y,z (cols 1-3) of |a-b| / max(|b|, 1)."""
    a, b = np.asarray(par_a)[:, :3], np.asarray(par_b)[:, :3]
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))
