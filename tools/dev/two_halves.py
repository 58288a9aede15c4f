"""development aid: does the chip do more when a step's chain of kernels runs as H independent part-chains beside each
other (the tail of one part's kernel filled by the other part's work)?  H engines in ONE process, each with 1/H of the
configs[2] particles and its window of the random stream, stepped in turn (every call is asynchronous), against one
engine with all of them.   usage: python3 tools/dev/two_halves.py [H ...]   (default: 1 2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import ltrans_amd
from ltrans_amd import roms_synth as rs, sharding

turb = os.environ.get("TH_TURB", "1") == "1"
grid = os.environ.get("TH_GRID", "512x256x30")
total = int(os.environ.get("TH_N", "1000000"))
vi, uj, us = (int(v) for v in grid.split("x"))
g, geo, bounds = bench.build_world(vi, uj, us, False)
rec = {}


def records(r):
    if r not in rec:
        rec[r] = rs.make_record(g, geo, r, with_kh=turb)
    return rec[r]


def run(H, steps=40, warm=10):
    engines = []
    sizes = sharding.shard_sizes(total, H)
    for h in range(H):
        parts = bench.make_parts(g, geo, bounds, sizes[h], 12345 + h, False)
        t = 1 if turb else 0
        cfg = ltrans_amd.LtConfig(numpar=sizes[h], us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, HTurbOn=t, VTurbOn=t,
                                  ConstantHTurb=1.0)
        e, r = bench.make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records)
        if turb:
            e.set_rng_window(int(sum(sizes[:h])), int(sum(sizes)), persistent=True)
        engines.append((e, bench.Stepper(e, r, cfg)))

    def step_all():
        for e, st in engines:
            ix = st.next_times()
            e.update_particles(st.ex, ix, st.p, st.it)

    def fence():
        for e, _ in engines:
            e.sync()

    for _ in range(warm):
        step_all()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        if engines[0][1].needs_rotation():
            break
        step_all()
    fence()
    dt = (time.perf_counter() - t0) / steps
    act = sum(e.counters()["active"] for e, _ in engines)
    print("H = %d: %.4f ms per step of %d particles (active %d)" % (H, dt * 1e3, total, act), flush=True)
    for e, _ in engines:
        e.close()


for H in ([int(a) for a in sys.argv[1:]] or [1, 2]):
    run(H, steps=18, warm=10)
