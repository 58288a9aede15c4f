"""development aid: wave cycles per section of k_rk (library built with -DLT_RK_TIMING)"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LT_HIP_LIBRARY"] = os.path.join(ROOT, "build", "libltrans_hip_timing.so")
import numpy as np
import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun
vi, uj, us = 512, 256, 30
n = 1_000_000
g = rs.make_grid(vi, uj, us); geo = rs.build_geometry(g); bounds = rs.create_bounds(g)
parts = rs.make_particles(g, geo, bounds, n, seed=12345)
cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1)
e = ltrans_amd.Engine(cfg)
e.set_grid(geo); e.set_bounds(bounds)
e.set_particles(**parts, **rs.behaviour_state(cfg, n))
rec = {}
run = LtransRun(e, cfg.dt, cfg.idt, lambda r: rec.setdefault(r, rs.make_record(g, geo, r, with_kh=False)))
run.init_hydro()
ex = run.external_times(2)
lib = e.lib
out = (C.c_ulonglong * 16)()
LATE = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # external steps to run first (30 internal steps each)
p = 2
for _ in range(LATE):
    run.run_external_timestep(p); p += 1
if LATE:
    run.p = p; run.update_hydro(); ex = run.external_times(p)
for it in range(1, 6):
    e.update_particles(ex, run.internal_times(ex, it), p, it)
e.sync(); lib.lt_diag(out)
K = 10
for it in range(6, 6 + K):
    e.update_particles(ex, run.internal_times(ex, it), p, it)
e.sync(); lib.lt_diag(out)
v = [int(x) for x in out]
names = ["0 regime+windows", "1 var setup (weights, nodes)", "2 X,Y,YPC1,select", "3 screen+queue+hval", "4 pass B", "5 pass C + stage tail"]
tot = sum(v[:6])
nwaves = (n + 63) // 64
for i in range(6):
    print(f"{names[i]:32s} {v[i]/K/nwaves:10.0f} cyc/wave/step  {100.0*v[i]/tot:5.1f} %")
print("total cyc/wave/step", tot / K / nwaves, " queued eval intervals/particle-step", v[6] / K / n, " regime>=3 lanes per particle-step (of 12)", (v[7] & 0xffffffff) / K / n, " var setups (of 12 per wave-step) with such a lane", (v[7] >> 32) / K / nwaves)
