"""development aid: wave cycles per part of k_post (library built with -DLT_RK_TIMING -DLT_POST_TIMING;
LT_DIAG_ENGINE_UNIT_ONLY=1), for variants of the configs[4] workload on the 1024x512x50 grid without turbulence:
LT_HIP_LIBRARY=build/lib_ptim.so LT_DIAG_ENGINE_UNIT_ONLY=1 python3 tools/dev/post_sections.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import ltrans_amd
from ltrans_amd import roms_synth as rs
n = 1_250_000
names = ["behaviour + new position", "intersect_reflect loop", "mbounds", "ibounds", "closing setEle (wave)", "settlement + commit"]
for label, reflect, behavior, settle, near in (("config4 (no islands, interior particles)", False, 0, False, False),
                                               ("islands + coast band", True, 0, False, True),
                                               ("config5", True, 6, True, True)):
    g, geo, bounds = bench.build_world(1024, 512, 50, reflect)
    parts = bench.make_parts(g, geo, bounds, n, 12345, near)
    drift = (-0.2, -0.15) if reflect else (0.0, 0.0)
    recs = {}
    cfg = ltrans_amd.LtConfig(numpar=n, us=50, ws=51, dt=3600, idt=120, ErrorFlag=1, **bench.behaviour_cfg(behavior, settle))
    e, run = bench.make_engine(ltrans_amd, cfg, g, geo, bounds, parts, lambda r: recs.setdefault(r, rs.make_record(g, geo, r, with_kh=False, drift=drift, with_st=False)))
    ex = run.external_times(2)
    out = (C.c_ulonglong * 16)()
    for it in range(1, 4): e.update_particles(ex, run.internal_times(ex, it), 2, it)
    e.sync(); e.lib.lt_diag(out)
    for it in range(4, 12): e.update_particles(ex, run.internal_times(ex, it), 2, it)
    e.sync(); e.lib.lt_diag(out)
    waves = max(out[7], 1)
    tot = sum(out[q] for q in range(6))
    print(label, ": %.0f wave cycles per wave in k_post" % (tot / waves))
    for q in range(6):
        print("   %-28s %9.0f cycles per wave  %5.1f %%" % (names[q], out[q] / waves, 100.0 * out[q] / max(tot, 1)))
    e.close()
