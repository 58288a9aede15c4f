#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the LTRANS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of update_particles() (LTRANS.f90:707) over the shard of
particles a rank holds: one internal time step of LTRANS.  Workload (N = 1):
BASELINE.json configs[1] -- 1M particles on a synthetic 512x256x30 ROMS grid,
RK4 advection only (HTurb / VTurb / behaviour off).  For N > 1 every rank holds
its own 1M-particle shard of the same grid (fields replicated, weak scaling,
SURVEY.md section 8e) and the ranks all-gather the new (x,y,z) over RCCL at the
end of each step.  Inputs are resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     algorithmic bytes / HIP-event kernel time vs the HBM peak
  "cpu_baseline": the CPU oracle (a single-thread C port of the reference path)
                  timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes one particle-step must touch,
# RK4 only, fp32 slab storage (coordinates / depth / angle stay fp64):
#   state 96 + grid statics 256 + zeta 4x3x4 = 48 + u,v,w 3 x (4 x 4 x 3 x 4 = 192) = 576
A_ADV_F32 = 96 + 256 + 48 + 576          # = 976 B
A_ADV_F64 = 96 + 256 + 96 + 1152         # = 1600 B
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--numpar", type=int, default=1_000_000, help="particles per GPU")
    ap.add_argument("--grid", type=str, default="512x256x30")
    ap.add_argument("--slab", choices=["f32", "f64"], default="f32")
    ap.add_argument("--turb", action="store_true",
                    help="BASELINE configs[2]: RK4 + HTurb + VTurb (MT19937 stream on device)")
    ap.add_argument("--behavior", type=int, default=0,
                    help="behaviour type 0-7 (BASELINE configs[4] uses 6 = sinking, with --settle)")
    ap.add_argument("--settle", action="store_true", help="settlement on habitat polygons + mortality")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hydro-timing", action="store_true",
                    help="skip the updateHydro cost measurement (set_slab vs slab pipeline)")
    ap.add_argument("--cpu-sample", type=int, default=40000,
                    help="particles of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=30)
    return ap.parse_args()


def pmc_traffic(tag, kernels, n, args):
    """HBM-side bytes per launch of the timed kernel(s) from the committed rocprofv3 PMC
    summary of THIS command (profiles/<tag>/pmc.json: FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, separate passes, see profiles/tools/).  Counters cannot be collected
    inside an ordinary run, so the number is the one measured for the default workload;
    any other workload reports null."""
    if n != 1_000_000 or args.grid != "512x256x30" or args.slab != "f32":
        return None, None
    p = os.path.join(ROOT, "profiles", tag, "pmc.json")
    try:
        d = json.load(open(p))
    except (OSError, ValueError):
        return None, None
    tot = 0.0
    valu = {}
    for kern in kernels:
        hit = [v for k, v in d.items() if ("::" + kern + "(" in k or "::" + kern + "<" in k)]
        if not hit or "fetch_bytes_per_launch" not in hit[0] or "write_bytes_per_launch" not in hit[0]:
            return None, None
        tot += hit[0]["fetch_bytes_per_launch"] + hit[0]["write_bytes_per_launch"]
        g = lambda c: hit[0].get(c, {}).get("mean_per_launch")  # noqa: E731
        if g("SQ_WAVES") and g("SQ_WAVE_CYCLES"):
            # what actually bounds these kernels: the fp64 VALU instruction stream
            valu[kern] = {"valu_instr_per_particle_step": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
                          "wave_cycles_valu_active": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
                          "wave_cycles_waiting": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")}
    pmc_traffic.valu = valu
    return tot, "profiles/%s/pmc.json" % tag


def valu_issue(valu, kernels, n, k_avg_s):
    """The bound that actually applies: VALU instruction issue.  PMC
    instruction count per particle-step (committed profile) x particles / live kernel time,
    against the MI355X vector rate of 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3e12
    lane-operations/s (the 78.6 TFLOP/s fp64 vector peak counts an FMA as two)."""
    if not valu or k_avg_s <= 0 or any(k not in valu for k in kernels):
        return None
    per = sum(valu[k]["valu_instr_per_particle_step"] for k in kernels)
    ach = per * n / k_avg_s / 1e12
    return {"achieved": ach, "peak": 39.3, "unit": "T lane-ops/s", "frac": ach / 39.3,
            "valu_instr_per_particle_step": per}


def build_inputs(vi, uj, us, numpar, seed):
    from ltrans_amd import roms_synth as rs
    g = rs.make_grid(vi, uj, us)
    geo = rs.build_geometry(g)
    bounds = rs.create_bounds(g)
    parts = rs.make_particles(g, geo, bounds, numpar, seed=seed)
    return g, geo, bounds, parts


def behaviour_cfg(args):
    """extra lt_config fields of the optional behaviour / settlement workload"""
    kw = dict(Behavior=args.behavior)
    if args.behavior:
        kw.update(swimstart=0.0, swimslow=0.002, swimfast=0.004, sink=-0.002, pediage=7200.0, deadage=90000.0,
                  Sgradient=0.3, Hswimspeed=0.3)
    if args.settle:
        kw.update(settlementon=1, holesExist=1, mortality=1, pediage=3600.0, deadage=36000.0)
    return kw


def make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records, **kw):
    from ltrans_amd import roms_synth as rs
    from ltrans_amd.driver import LtransRun
    e = ltrans_amd.Engine(cfg, **kw)
    e.set_grid(geo)
    e.set_bounds(bounds)
    if cfg.settlementon:
        Lx, Ly = (g["vi"] - 1) * g["dx"], (g["uj"] - 1) * g["dy"]
        polys = [(Lx * fx, Ly * fy, 0.03 * Lx, 14) for fx in (0.2, 0.4, 0.6, 0.8) for fy in (0.25, 0.5, 0.75)]
        e.set_habitat(rs.make_habitat(g, geo, polys, [(0, Lx * 0.2, Ly * 0.25, 0.008 * Lx, 8)]))
    e.set_particles(**parts, **rs.behaviour_state(cfg, cfg.numpar))
    run = LtransRun(e, cfg.dt, cfg.idt, lambda r: records(r))
    run.init_hydro()
    return e, run


def hydro_update_costs(e, run, rs, g, geo, args, state, one_step, fence):
    """What one updateHydro costs on this box, outside the throughput measurement
    (SURVEY.md 8f row 1): the synchronous lt_set_slab path (host waits, GPU idle) against
    the slab pipeline (lt_slab_prefetch beside the steps, lt_slab_commit at the rotation).
    ``exposed`` = wall of [prefetch, stepIT steps, commit] - wall of [stepIT steps]."""
    import numpy as np
    stepIT = run.dt // run.idt
    r0 = run.stepf
    masked = rs.make_record(g, geo, r0, with_kh=bool(args.turb))
    rom32 = rs.make_record(g, geo, r0 + 1, with_kh=bool(args.turb), raw=np.float32)
    rom64 = rs.make_record(g, geo, r0 + 2, with_kh=bool(args.turb), raw=np.float64)
    dims = (g["vi"], g["uj"], g["ui"], g["vj"])
    mb = lambda d: sum(v.nbytes for v in d.values() if v is not None) / 1e6  # noqa: E731

    def steps():
        for _ in range(stepIT):
            state["it"] = state["it"] % stepIT
            one_step()

    def rotated():
        return run.t_b % 3 + 1, run.t_c % 3 + 1, run.t_f % 3 + 1

    fence()
    t0 = time.perf_counter(); steps(); fence(); t_steps = time.perf_counter() - t0
    # synchronous path
    run.t_b, run.t_c, run.t_f = rotated()
    t0 = time.perf_counter()
    e.set_slab(run.t_f, masked["zeta"], masked["Uvel"], masked["Vvel"], masked["Wvel"], masked.get("Kh"),
               masked.get("salt"), masked.get("temp"))
    e.set_time_slots(run.t_b, run.t_c, run.t_f)
    fence()
    t_sync = time.perf_counter() - t0
    out = {"steps_per_external_step": stepIT, "steps_ms": t_steps * 1e3,
           "set_slab": {"host_mb": mb(masked), "ms": t_sync * 1e3}}
    # first use allocates the pinned staging buffers: not part of a steady-state update
    e.slab_prefetch(rom32, dims)
    run.t_b, run.t_c, run.t_f = rotated()
    e.slab_commit(run.t_b, run.t_c, run.t_f)
    for name, rom in (("pipeline_f32_host_arrays", rom32), ("pipeline_f64_host_arrays", rom64)):
        fence()
        t0 = time.perf_counter()
        e.slab_prefetch(rom, dims)
        t_call = time.perf_counter() - t0
        steps()
        run.t_b, run.t_c, run.t_f = rotated()
        tc = time.perf_counter()
        e.slab_commit(run.t_b, run.t_c, run.t_f)
        t_commit = time.perf_counter() - tc
        fence()
        t_all = time.perf_counter() - t0
        out[name] = {"host_mb": mb(rom), "prefetch_call_ms": t_call * 1e3, "commit_call_ms": t_commit * 1e3,
                     "exposed_ms": (t_all - t_steps) * 1e3}
    return out


def cpu_baseline(ltrans_amd, args, vi, uj, us, g, geo, bounds, records):
    """Time the CPU oracle (kind "port": single-thread C restatement of the
    reference path, oracle/ltrans_oracle.c) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from ltrans_amd import roms_synth as rs
    n = args.cpu_sample
    parts = rs.make_particles(g, geo, bounds, n, seed=777)
    turb = 1 if args.turb else 0
    cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120,
                              ErrorFlag=1, HTurbOn=turb, VTurbOn=turb, **behaviour_cfg(args))
    e, run = make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records,
                         lib=oracle_lib.load_oracle(), prefix="lto_")
    ex = run.external_times(2)
    t0 = time.perf_counter()
    for it in range(1, args.cpu_steps + 1):
        e.update_particles(ex, run.internal_times(ex, it), 2, it)
    dt = time.perf_counter() - t0
    e.close()
    return {"value": n * args.cpu_steps / dt, "unit": "particle-steps/s", "cores": 1,
            "kind": "port",
            "sample": f"{n} particles x {args.cpu_steps} internal steps on the same "
                      f"{vi}x{uj}x{us} grid, {dt:.1f} s wall, {os.cpu_count()} host cores present"}


def main():
    args = parse()
    import numpy as np
    import torch
    import ltrans_amd
    from ltrans_amd import roms_synth as rs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # LT_BENCH_BACKEND=gloo (rehearsal on a box with fewer GPUs than ranks: ranks share the
    # cards, the all-gather goes through host memory) -- the driver's runs use RCCL
    backend = os.environ.get("LT_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = (local_rank % ndev) if (world > 1 and backend != "nccl") else (local_rank if world > 1 else 0)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    vi, uj, us = (int(v) for v in args.grid.split("x"))
    n = args.numpar
    g, geo, bounds, parts = build_inputs(vi, uj, us, n, seed=12345 + rank)
    rec_cache = {}

    def records(r):
        if r not in rec_cache:
            rec_cache[r] = rs.make_record(g, geo, r, with_kh=bool(args.turb))
        return rec_cache[r]

    turb = 1 if args.turb else 0
    cfg = ltrans_amd.LtConfig(numpar=n, us=us, ws=us + 1, dt=3600, idt=120,
                              ErrorFlag=1, device=dev.index, HTurbOn=turb, VTurbOn=turb, **behaviour_cfg(args),
                              slab_storage=(ltrans_amd.abi.LT_SLAB_F32 if args.slab == "f32"
                                            else ltrans_amd.abi.LT_SLAB_F64))
    e, run = make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records)

    # N > 1: the end-of-step all-gather of (x,y,z) runs on its own stream and overlaps the
    # next step's kernels; the packed positions are double-buffered so a step never
    # overwrites a buffer RCCL may still be reading.
    send = recv = gather_stream = None
    send_done = [None, None]
    if world > 1:
        send = [torch.empty(3 * n, dtype=torch.float64, device=dev) for _ in range(2)]
        recv = torch.empty(3 * n * world, dtype=torch.float64, device=dev)
        gather_stream = torch.cuda.Stream(device=dev)

    p = 2                      # second external step: plain (b,c,f) time interpolation
    ex = run.external_times(p)
    stepIT = cfg.dt // cfg.idt
    state = {"it": 0, "p": p, "ex": ex, "nstep": 0}

    def needs_rotation():
        return state["it"] + 1 > stepIT

    def rotate():
        """next external step: updateHydro = rotate slots + upload ONE record.  Kept
        outside the timed region (slab upload is reported separately)."""
        state["p"] += 1
        state["it"] = 0
        run.p = state["p"]
        run.update_hydro()
        state["ex"] = run.external_times(state["p"])

    def one_step():
        state["it"] += 1
        ix = run.internal_times(state["ex"], state["it"])
        if world > 1 and (args.turb or 1 <= args.behavior <= 5):
            # every rank's place in the single serial random stream of this step
            from ltrans_amd import sharding
            if 1 <= args.behavior <= 5:   # data-dependent number of deviates: count draws
                mine = e.count_draws(state["ex"], ix, state["p"], state["it"])
                before, total = sharding.rng_window(dist, torch, mine, world, rank, device=coll_dev)
                e.set_rng_window_draws(before, total)
            else:
                before, total = sharding.rng_window(dist, torch, e.count_active(ix), world, rank, device=coll_dev)
                e.set_rng_window(before, total)
        e.update_particles(state["ex"], ix, state["p"], state["it"])
        if world > 1:
            b = state["nstep"] & 1
            state["nstep"] += 1
            if send_done[b] is not None:
                send_done[b].synchronize()      # the all-gather that last read this buffer
            e.pack_positions(send[b].data_ptr())
            e.sync()                            # engine stream: step + pack complete
            with torch.cuda.stream(gather_stream):
                if backend == "nccl":
                    dist.all_gather_into_tensor(recv, send[b])
                else:
                    parts_ = [torch.empty(3 * n, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts_, send[b].cpu())
                    recv.copy_(torch.cat(parts_))
                send_done[b] = torch.cuda.Event()
                send_done[b].record(gather_stream)

    def fence():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        if needs_rotation():
            rotate()
        one_step()
    fence()
    e.profile_begin(args.steps)
    wall = 0.0
    upload_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if needs_rotation():
            fence()
            wall += time.perf_counter() - t0
            tu = time.perf_counter()
            rotate()
            fence()
            upload_s += time.perf_counter() - tu
            t0 = time.perf_counter()
        one_step()
    fence()
    wall += time.perf_counter() - t0
    k_ms, k_n = e.profile_end()
    stats = e.counters()

    wall_t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    active = stats["active"]
    total_particle_steps = float(n) * world * args.steps
    value = total_particle_steps / wall
    A = A_ADV_F32 if args.slab == "f32" else A_ADV_F64
    if args.turb:
        # + Kh 4 nodes x ws x 3 slots, + the materialised stream (124 draws
        # of 4 B written once and read once)
        ws_ = us + 1
        A += (48 if args.slab == "f32" else 96) * ws_ + 124 * 4 * 2
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = (A * n / k_avg_s) / 1e9 if k_avg_s > 0 else 0.0
    kernels = ["k_vt_table", "k_vt_sigs", "k_vt_walk"] if args.turb else ["k_rk"]
    traffic, traffic_src = pmc_traffic("r01_turb" if args.turb else "r01_adv", kernels, n, args)

    out = {
        "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} particles per GPU, synthetic {vi}x{uj}x{us} ROMS grid, "
                               + ("RK4 + HTurb + VTurb (MT19937 stream on device), idt=120 s" if args.turb
                                  else "RK4 advection only (HTurb/VTurb off), idt=120 s")
                               + (f", behaviour {args.behavior}" if args.behavior else ", behavior off")
                               + (", settlement + mortality" if args.settle else "")
                               + (", RCCL all-gather of (x,y,z) per step" if world > 1 else ""),
                   "particles_per_gpu": n, "grid": args.grid, "slab_storage": args.slab,
                   "active_last_step": active, "errors_last_step": stats["errors"],
                   "sigerr_last_step": stats["sigerr"],
                   "slab_rotation_s_excluded": upload_s},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "kernel": "+".join(kernels), "kernel_ms_avg": k_avg_s * 1e3,
                     "valu_pmc": getattr(pmc_traffic, "valu", None),
                     "valu_issue": valu_issue(getattr(pmc_traffic, "valu", None), kernels, n, k_avg_s),
                     "algorithmic_bytes_per_launch": A * n,
                     "algorithmic_bytes_per_particle_step": A},
    }
    if world == 1 and not args.no_hydro_timing:
        out["config"]["hydro_update"] = hydro_update_costs(e, run, rs, g, geo, args, state, one_step, fence)
        # the print step: lon/lat/status on the device + download of the
        # writers' columns, against downloading the whole particle state
        fence()
        e.get_output()
        t0 = time.perf_counter(); e.get_output(); t_out = time.perf_counter() - t0
        t0 = time.perf_counter(); e.get_particles(); t_all = time.perf_counter() - t0
        out["config"]["print_step"] = {"lt_get_output_ms": t_out * 1e3, "lt_get_particles_ms": t_all * 1e3}
        # init-time geometry: setEle_all for this shard's particles and the three
        # adjacency tables on the device, checked against the host-side construction of the inputs
        px, py = parts["par"][:, 0].copy(), parts["par"][:, 1].copy()
        e.locate_particles(px[:1024], py[:1024])
        t0 = time.perf_counter(); er, eu, ev, lerr, _ = e.locate_particles(px, py); t_loc = time.perf_counter() - t0
        t0 = time.perf_counter()
        adj = [e.build_adjacency(geo[k], row) for k, row in (("RE", g["vi"]), ("UE", g["ui"]), ("VE", g["vi"]))]
        t_adj = time.perf_counter() - t0
        same = (lerr == 0 and np.array_equal(er, parts["P_r_element"]) and np.array_equal(eu, parts["P_u_element"])
                and np.array_equal(ev, parts["P_v_element"]) and np.array_equal(adj[0], geo["r_Adjacent"])
                and np.array_equal(adj[1], geo["u_Adjacent"]) and np.array_equal(adj[2], geo["v_Adjacent"]))
        out["config"]["init_geometry"] = {"lt_locate_particles_ms": t_loc * 1e3, "lt_build_adjacency_ms": t_adj * 1e3,
                                          "matches_host_construction": bool(same)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ltrans_amd, args, vi, uj, us, g, geo, bounds, records)
    elif rank == 0:
        out["cpu_baseline"] = None
    e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
