"""development aid: k_rk and wall time of the first steps after set-up, with and without a busy GPU before them
(python3 tools/dev/step_trend.py [milliseconds of fp32 matrix products first]): the clock ramp bench.py pre-warms away"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
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
NS = 28
e.sync()
if len(sys.argv) > 1:
    ms = float(sys.argv[1])
    kind = sys.argv[2] if len(sys.argv) > 2 else "matmul"
    if kind == "matmul":
        a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
    else:
        a = torch.ones(1 << 26, device="cuda")
    if kind == "matmul": c = a @ b      # (the first product loads the library: not counted)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(10):
            if kind == "matmul": c = a @ b
            else: a.mul_(1.0000001)
        torch.cuda.synchronize()
    print("pre-warmed the clocks for %.0f ms with %s" % ((time.perf_counter() - t0) * 1e3, kind))
walls = []; ks = []
for it in range(1, NS + 1):
    e.profile_begin(1)
    t_prev = time.perf_counter()
    e.update_particles(ex, run.internal_times(ex, it), 2, it)
    e.sync(); t = time.perf_counter(); walls.append((t - t_prev) * 1e3)
    ks.append(e.profile_end()[0])
print("k_rk ms per step :", " ".join("%.3f" % k for k in ks))
print("wall ms per step :", " ".join("%.3f" % w for w in walls))
