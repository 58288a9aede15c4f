"""development aid: what the waves of k_vt_walk (the listed walks) spend their sub-steps on (library built with
-DLT_RK_TIMING: tools/dev/mk.sh timing -DLT_RK_TIMING; cp build/lib_timing.so build/libltrans_hip_timing.so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LT_DIAG_ENGINE_UNIT_ONLY"] = "1"
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
v = [int(x) / K for x in out]
w = v[0]
print("working waves per step %.0f" % w)
print("iterations of a wave (its longest lane)      %.1f" % (v[1] / w))
print("lane-iterations per wave                     %.0f  (%.0f %% of 64 x iterations)" % (v[2] / w, 100 * v[2] / (64 * v[1])))
print("lane-iterations on a tension interval        %.0f  (%.1f %% of the lane-iterations)" % (v[3] / w, 100 * v[3] / v[2]))
print("iterations with any lane on one              %.1f  rational form %.1f  exponential form %.1f" % (v[4] / w, v[5] / w, v[6] / w))
print("most tension sub-steps of one lane of a wave %.1f" % (v[7] / w))
print("cycles per wave %.0f, per iteration %.0f; the longest wave of the %d steps %.0f" % (v[14] / w, v[14] / v[1], K, v[15] * K))
if os.environ.get("LT_VT_ADV_STATS"):   # library built with -DLT_VT_ADV_STATS as well: slots 5..7 are k_vt_table's
    print("k_vt_table: waves %.0f, executions of the segment advance per wave %.1f (of %d knot x slot), lanes in one %.1f" % (
        v[5], v[6] / v[5], 3 * 4 * (us + 1), v[7] / v[6]))
