"""development aid: start / end of every wave of one k_rk launch (library built with -DLT_RK_WAVETIME:
tools/dev/mk.sh wavetime -DLT_RK_WAVETIME): wave durations, waves in flight over the launch, the shader clock"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LT_HIP_LIBRARY"] = os.path.join(ROOT, "build", "lib_wavetime.so")
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
for it in range(1, 16):
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
e.sync()
out = (C.c_ulonglong * 40000)()
lib.lt_diag_wavetime(out)
nw = (n + 63) // 64
t = np.array(out[:2 * nw], dtype=np.int64).reshape(nw, 2)
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us
dur = end - start
print("waves", nw, "kernel span us", end.max(), "wave duration us: mean %.1f p5 %.1f p50 %.1f p95 %.1f max %.1f" % (
    dur.mean(), *np.percentile(dur, [5, 50, 95]), dur.max()))
vb = np.arange(nw)
# waves in flight over time
edges = np.linspace(0, end.max(), 41)
mid = 0.5 * (edges[1:] + edges[:-1])
infl = [(np.sum((start <= m) & (end > m))) for m in mid]
print("waves in flight at 40 instants:", infl)
print("sum of wave durations / span = average concurrency %.0f" % (dur.sum() / end.max()))
# duration of the chunks by start time, and when each eighth (XCD share) was finished
per = nw // 8
bins = np.linspace(0, start.max() + 1e-9, 11)
for b in range(10):
    m = (start >= bins[b]) & (start < bins[b + 1])
    if m.any():
        print("  chunks started in [%5.0f, %5.0f) us: %5d  mean duration %.1f  p95 %.1f" % (bins[b], bins[b + 1], m.sum(), dur[m].mean(), np.percentile(dur[m], 95)))
print("  last end per eighth:", [round(float(end[x * per:(x + 1) * per].max()), 1) for x in range(8)])
print("  mean duration per eighth:", [round(float(dur[x * per:(x + 1) * per].mean()), 1) for x in range(8)])
cyc = (C.c_ulonglong * 20000)()
lib.lt_diag_wavecyc(cyc)
cy = np.array(cyc[:nw], dtype=np.float64)
print("shader clock while a chunk runs: mean %.3f GHz (cycle counter / 100 MHz clock), p5 %.3f p95 %.3f" % (
    (cy / (dur * 1e3)).mean(), *np.percentile(cy / (dur * 1e3), [5, 95])))
