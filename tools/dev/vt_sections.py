"""development aid: wave cycles per section of a knot of k_vt_table (library built with -DLT_RK_TIMING:
tools/dev/mk.sh timing -DLT_RK_TIMING; cp build/lib_timing.so build/libltrans_hip_timing.so)"""
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
for it in range(1, 4):
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
e.sync(); e.lib.lt_diag(out)
K = 5
for it in range(4, 4 + K):
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
e.sync(); e.lib.lt_diag(out)
v = [int(x) for x in out][8:14]
names = ["0 window average, newx, mid points", "1 polintd x2, clamp, Simpson", "2 S, YPC1", "3 screening, task list",
         "4 record stores, state", "5 window slide: newy(k+8), segment advance; loop"]
tot = sum(v); nw = (n + 63) // 64; knots = 4 * (us + 1)
for i in range(6):
    print(f"{names[i]:52s} {v[i]/K/nw/knots:8.0f} cyc/knot  {100.0*v[i]/tot:5.1f} %")
print("total cycles per wave and knot", tot / K / nw / knots)
