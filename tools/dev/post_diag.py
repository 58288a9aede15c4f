"""development aid: k_post's time (HIP events around lt_step minus ...) is not separable from outside; this prints the
step's counters and wall time for variants of the configs[4] workload on the 1024x512x50 grid:
python3 tools/dev/post_diag.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import ltrans_amd
from ltrans_amd import roms_synth as rs
n = 1_250_000
for label, reflect, behavior, settle, near in (("config4 (no islands, interior particles)", False, 0, False, False),
                                               ("islands, interior particles", True, 0, False, False),
                                               ("islands + coast band", True, 0, False, True),
                                               ("islands + coast band + behaviour 6", True, 6, False, True),
                                               ("config5: + settlement + mortality", True, 6, True, True)):
    g, geo, bounds = bench.build_world(1024, 512, 50, reflect)
    parts = bench.make_parts(g, geo, bounds, n, 12345, near)
    drift = (-0.2, -0.15) if reflect else (0.0, 0.0)
    recs = {}
    cfg = ltrans_amd.LtConfig(numpar=n, us=50, ws=51, dt=3600, idt=120, ErrorFlag=1, HTurbOn=0, VTurbOn=0, ConstantHTurb=1.0,
                              **bench.behaviour_cfg(behavior, settle))
    e, run = bench.make_engine(ltrans_amd, cfg, g, geo, bounds, parts, lambda r: recs.setdefault(r, rs.make_record(g, geo, r, with_kh=False, drift=drift, with_st=False)))
    ex = run.external_times(2)
    for it in range(1, 4): e.update_particles(ex, run.internal_times(ex, it), 2, it)
    e.sync(); t0 = time.perf_counter()
    for it in range(4, 20): e.update_particles(ex, run.internal_times(ex, it), 2, it)
    e.sync(); dt = (time.perf_counter() - t0) / 16
    c = e.counters()
    print("%-44s %.3f ms/step (advection only)  brute %d pip_full %d errors %d segments %d" % (label, dt * 1e3, c["brute_bounds"], c["pip_full"], c["errors"], bounds["nbounds"]))
    e.close()
