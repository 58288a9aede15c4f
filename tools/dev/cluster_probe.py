"""development aid: k_rk time with all particles inside a few cells (every gather hits the
caches) against the uniform distribution: how much of the kernel is memory latency"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun
vi, uj, us = 512, 256, 30
n = 1_000_000
g = rs.make_grid(vi, uj, us); geo = rs.build_geometry(g); bounds = rs.create_bounds(g)
rec = {}
def run(parts, tag):
    cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1)
    e = ltrans_amd.Engine(cfg)
    e.set_grid(geo); e.set_bounds(bounds)
    e.set_particles(**parts, **rs.behaviour_state(cfg, n))
    r = LtransRun(e, cfg.dt, cfg.idt, lambda k: rec.setdefault(k, rs.make_record(g, geo, k, with_kh=False)))
    r.init_hydro()
    ex = r.external_times(2)
    for it in range(1, 6):
        e.update_particles(ex, r.internal_times(ex, it), 2, it)
    e.sync()
    e.profile_begin(20)
    for it in range(6, 26):
        e.update_particles(ex, r.internal_times(ex, it), 2, it)
    ms, k = e.profile_end()
    print(tag, "k_rk ms", ms / k, e.counters())
    e.close()
parts = rs.make_particles(g, geo, bounds, n, seed=12345)
run(parts, "uniform")
for cells in (64, 16, 4):
    rng = np.random.default_rng(1)
    x = (200.0 + rng.uniform(0, cells, n)) * g["dx"]
    y = (100.0 + rng.uniform(0, cells, n)) * g["dy"]
    p2 = {k: v.copy() for k, v in parts.items()}
    p2["par"][:, 0] = x; p2["par"][:, 1] = y
    p2["par"][:, 3:5] = p2["par"][:, 0:2]; p2["par"][:, 6:8] = p2["par"][:, 0:2]
    for kind, key in (("r", "P_r_element"), ("u", "P_u_element"), ("v", "P_v_element")):
        p2[key] = rs.locate_elements(g, geo, x, y, kind)
    h = np.interp(0, [0], [0])
    gi = np.floor(x / g["dx"]).astype(int); gj = np.floor(y / g["dy"]).astype(int)
    hloc = np.minimum.reduce([g["h"][gi, gj], g["h"][gi + 1, gj], g["h"][gi, gj + 1], g["h"][gi + 1, gj + 1]])
    p2["par"][:, 2] = -rng.uniform(0.5 / hloc, 0.9, n) * hloc
    p2["par"][:, 5] = p2["par"][:, 2]; p2["par"][:, 8] = p2["par"][:, 2]
    run(p2, "cluster %dx%d cells" % (cells, cells))
