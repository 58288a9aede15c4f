"""development aid: how far (in spline intervals of the 4*ws-1) the walks of VTurb go from the interval of their start
depth -- what a table restricted to a window of knots round the start would have to cover (library built with
-DLT_RK_TIMING -DLT_VT_EXCURSION: tools/dev/mk.sh timing -DLT_RK_TIMING -DLT_VT_EXCURSION;
cp build/lib_timing.so build/libltrans_hip_timing.so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LT_HIP_LIBRARY"] = os.path.join(ROOT, "build", "libltrans_hip_timing.so")
import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun
vi, uj, us = 512, 256, 30
n = 1_000_000
g = rs.make_grid(vi, uj, us); geo = rs.build_geometry(g); bounds = rs.create_bounds(g)
parts = rs.make_particles(g, geo, bounds, n, seed=12345)
cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0)
e = ltrans_amd.Engine(cfg)
e.set_grid(geo); e.set_bounds(bounds)
e.set_particles(**parts, **rs.behaviour_state(cfg, n))
rec = {}
run = LtransRun(e, cfg.dt, cfg.idt, lambda r: rec.setdefault(r, rs.make_record(g, geo, r, with_kh=True)))
run.init_hydro()
ex = run.external_times(2)
out = (C.c_ulonglong * 16)()
e.update_particles(ex, run.internal_times(ex, 1), 2, 1)
e.sync(); e.lib.lt_diag(out)
K = 10
for it in range(2, 2 + K):
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
e.sync(); e.lib.lt_diag(out)
v = [int(x) for x in out]
tot = sum(v)
print("walk segments counted (a walk that met a tension factor counts twice: before and after): %d over %d steps" % (tot, K))
acc = 0
for i, c in enumerate(v):
    acc += c
    print("  excursion %s%2d intervals: %9d  (%.4f %% go further)" % (">=" if i == 15 else "  ", i, c, 100.0 * (tot - acc) / tot))
