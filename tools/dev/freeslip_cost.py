"""development aid: what FreeSlip costs.  configs[4]'s world (islands, a fifth of the particles within 120 m of a coast) on
the one-GPU grid, advection only, FreeSlip off / on: ms per step and the length of k_rk's slow list (the particle-steps
whose element has a land node go to k_rk_slow's straight transcription).   usage: python3 tools/dev/freeslip_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import ltrans_amd
from ltrans_amd import roms_synth as rs

vi, uj, us = 512, 256, 30
n = int(os.environ.get("FS_N", "1000000"))
g, geo, bounds = bench.build_world(vi, uj, us, True)
parts = bench.make_parts(g, geo, bounds, n, 12345, True)
rec = {}


def records(r):
    if r not in rec:
        rec[r] = rs.make_record(g, geo, r, with_kh=False, drift=(-0.2, -0.15))
    return rec[r]


for fs in (0, 1):
    cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, FreeSlip=fs)
    e, run = bench.make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records)
    st = bench.Stepper(e, run, cfg)
    for _ in range(8):
        e.update_particles(st.ex, st.next_times(), st.p, st.it)
    e.sync()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        e.update_particles(st.ex, st.next_times(), st.p, st.it)
    e.sync()
    dt = (time.perf_counter() - t0) / K
    c = e.counters()
    print("FreeSlip %d: %.4f ms per step, slow list of the last step %s of %d active" % (fs, dt * 1e3, c.get("rk_slow", c.get("slow")), c["active"]), flush=True)
    e.close()
