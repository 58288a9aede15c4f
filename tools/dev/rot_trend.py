"""development aid: wall time of every step (host fence after each) through slab rotations by the pipeline and re-sorts:
python3 tools/dev/rot_trend.py [--turb] [--steps N] [--nofence K]   (--nofence K: fence only every K steps)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun
ap = argparse.ArgumentParser(); ap.add_argument("--turb", action="store_true"); ap.add_argument("--steps", type=int, default=75)
ap.add_argument("--nofence", type=int, default=1); ap.add_argument("--sync-rot", action="store_true")
a = ap.parse_args()
vi, uj, us, n = 512, 256, 30, 1_000_000
g = rs.make_grid(vi, uj, us); geo = rs.build_geometry(g); bounds = rs.create_bounds(g)
parts = rs.make_particles(g, geo, bounds, n, seed=12345)
t = 1 if a.turb else 0
cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, HTurbOn=t, VTurbOn=t, ConstantHTurb=1.0)
e = ltrans_amd.Engine(cfg); e.set_grid(geo); e.set_bounds(bounds); e.set_particles(**parts, **rs.behaviour_state(cfg, n))
nrot = a.steps // 30 + 2
masked = {r: rs.make_record(g, geo, r, with_kh=bool(t), with_st=False) for r in range(3)}
roms = {r: rs.make_record(g, geo, r, with_kh=bool(t), with_st=False, raw=np.float32) for r in range(3, 3 + nrot)}
run = LtransRun(e, cfg.dt, cfg.idt, lambda r: masked[r], rom_fn=None if a.sync_rot else (lambda r: roms[r]), dims=(g["vi"], g["uj"], g["ui"], g["vj"]))
if a.sync_rot:
    for r in roms: masked[r] = rs.make_record(g, geo, r, with_kh=bool(t), with_st=False)
run.init_hydro()
if t: e.set_rng_window(0, n, persistent=True)
p, it = 2, 0
ex = run.external_times(p)
run.p = p
run.prefetch_next()
out = []
e.sync(); t0 = time.perf_counter(); tl = t0
for s in range(a.steps):
    tag = ""
    if it + 1 > 30:
        p += 1; it = 0; run.p = p
        tr = time.perf_counter(); run.update_hydro(); run.prefetch_next(); tag = "R(%.2f)" % ((time.perf_counter() - tr) * 1e3)
        ex = run.external_times(p)
    it += 1
    e.update_particles(ex, run.internal_times(ex, it), p, it)
    if (s + 1) % a.nofence == 0:
        e.sync(); tn = time.perf_counter(); out.append("%s%.2f" % (tag, (tn - tl) * 1e3 / a.nofence)); tl = tn
    elif tag:
        out.append(tag)
e.sync()
print("ms per step (R(host ms of the rotation calls) marks a slab rotation):"); print(" ".join(out))
print("total %.2f ms for %d steps = %.3f ms/step; resorts %d" % ((time.perf_counter() - t0) * 1e3, a.steps, (time.perf_counter() - t0) * 1e3 / a.steps, e.counters()["resorts"]))
