import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun
vi, uj, us = 512, 256, 30; n = 1_000_000
g = rs.make_grid(vi, uj, us); geo = rs.build_geometry(g); bounds = rs.create_bounds(g)
parts = rs.make_particles(g, geo, bounds, n, seed=12345)
cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0)
e = ltrans_amd.Engine(cfg); e.set_grid(geo); e.set_bounds(bounds); e.set_particles(**parts, **rs.behaviour_state(cfg, n))
rec = {}
run = LtransRun(e, cfg.dt, cfg.idt, lambda r: rec.setdefault(r, rs.make_record(g, geo, r, with_kh=True))); run.init_hydro()
ex = run.external_times(2)
for it in range(1, 4):
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
c = e.counters(); print(c, "queued per particle", c["vt_queued"] / n, "rounds per wave", c["vt_rounds"] / (n / 64), "tasks per round", c["vt_queued"] / max(c["vt_rounds"], 1))
