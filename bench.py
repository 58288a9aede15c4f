#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the LTRANS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of update_particles() (LTRANS.f90:707) over the shard of
particles a rank holds: one internal time step of LTRANS.

  N = 1   BASELINE.json configs[1]: 1M particles on a synthetic 512x256x30 ROMS grid, RK4
          advection only (the configuration the metric is quoted on).  The same line carries,
          under config.turbulence, configs[2] (the same + HTurb + VTurb with the device
          MT19937 stream) timed for the same number of steps.
  N > 1   the same configuration PER GPU (1M particles each, fields replicated, particles sharded
          by contiguous index block): WEAK scaling, so that the per-N
          values form one series with the N = 1 line.  After every step the ranks all-gather the
          new (x,y,z) over RCCL, on a stream of its own beside the next step.  The same line
          carries, under config.configs3_strong, BASELINE.json configs[3] -- 10M particles in
          total on a 1024x512x50 grid, RK4 + HTurb + VTurb, STRONG-scaled over the same ranks --
          with rank 0's run of all its particles on one GPU (the one-GPU point of that workload).
          --workload picks one configuration explicitly (strong scaling unless --weak).

When started without a launcher (`python bench.py --gpus N`, N > 1) this process only starts
N children (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's
line and exits non-zero if any child fails; it never touches the GPU itself.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks exist
already; a WORLD_SIZE that differs from --gpus is an error.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     algorithmic bytes / HIP-event kernel time vs the HBM peak
  "cpu_baseline": the CPU oracle (a single-thread C port of the reference path)
                  timed on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
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
PROFILE_ADV, PROFILE_TURB = "r05_adv", "r05_turb"   # profiles/<tag>/pmc.json of the default workloads

WORKLOADS = {
    # name: (grid, total particles, turbulence, reflect/behaviour-6/settlement)
    "config2": ("512x256x30", 1_000_000, False, False),
    "config3": ("512x256x30", 1_000_000, True, False),
    "config4": ("1024x512x50", 10_000_000, True, False),
    "config5": ("1024x512x50", 10_000_000, True, True),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-ms", type=float, default=40.0,
                    help="busy the GPU this long before the warm-up steps so that the clocks are up (0: off)")
    ap.add_argument("--workload", choices=["auto"] + sorted(WORKLOADS), default="auto",
                    help="auto = config2 (the configuration the metric is quoted on) per GPU: on several GPUs weak scaling, "
                         "with configs[3] strong-scaled timed beside it (config.configs3_strong)")
    ap.add_argument("--weak", action="store_true", help="N > 1, explicit --workload: --numpar particles PER RANK (weak scaling)")
    ap.add_argument("--no-configs3-line", action="store_true",
                    help="N > 1, --workload auto: skip the embedded configs[3] (10M particles, 1024x512x50, strong scaling) measurement")
    ap.add_argument("--configs3-grid", type=str, default=None, help="grid of the embedded configs[3] line (rehearsals)")
    ap.add_argument("--configs3-numpar", type=int, default=None, help="total particles of the embedded configs[3] line")
    ap.add_argument("--numpar", type=int, default=None, help="particles (per GPU with --weak, else in total)")
    ap.add_argument("--grid", type=str, default=None)
    ap.add_argument("--slab", choices=["f32", "f64"], default="f32")
    ap.add_argument("--turb", action="store_true", help="= --workload config3 on one GPU")
    ap.add_argument("--behavior", type=int, default=None,
                    help="behaviour type 0-7 (BASELINE configs[4] uses 6 = sinking, with --settle)")
    ap.add_argument("--settle", action="store_true", help="settlement on habitat polygons + mortality")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hydro-timing", action="store_true",
                    help="skip the updateHydro / print-step / init-geometry cost measurements")
    ap.add_argument("--no-turbulence-line", action="store_true",
                    help="N = 1, config2: skip the embedded configs[2] measurement (config.turbulence)")
    ap.add_argument("--emulate-rank", default=None, metavar="R:W",
                    help="one process does rank R's part of a W-rank run of the workload (its shard, its static window "
                         "of the random stream; no all-gather): what one GPU of the N-rank job computes per step")
    ap.add_argument("--no-single-ref", action="store_true",
                    help="N > 1, strong scaling: skip rank 0's run of the WHOLE workload on one GPU afterwards")
    ap.add_argument("--single-ref-budget", type=float, default=150.0,
                    help="seconds after which rank 0's single-GPU reference run is skipped (keeps --gpus 8 within minutes)")
    ap.add_argument("--sustained-steps", type=int, default=320,
                    help="N = 1 main line: after the timed window, this many further steps of the SAME engine timed as one "
                         "series with its re-sorts and slab rotations (through the pipeline) inside: config.sustained_* (0: off)")
    ap.add_argument("--cpu-sample", type=int, default=40000, help="particles of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--selftest-launch", choices=["ok", "fail"], default=None,
                    help="only exercise the launcher: ranks rendezvous over gloo on CPU tensors and print a line")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(args):
    """--gpus N without a launcher: start the N ranks.  Nothing here imports torch or touches HIP."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    out0 = b""
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            p = procs[r]
            try:
                if r == 0:
                    o, _ = p.communicate(timeout=0.2)
                    out0 += o or b""
                else:
                    p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if p.returncode != 0:
                failed = (r, p.returncode)
                break
    if failed is not None:
        for r in pending:                       # the exact children this process started
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write("bench.py: rank %d exited with code %d\n" % failed)
        sys.stdout.write(out0.decode(errors="replace"))
        return 1
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return 0


_LINE_FD = None


def claim_stdout():
    """The process's stdout carries ONE thing: rank 0's JSON line.  Whatever else a rank or a library under it writes
    to file descriptor 1 -- gloo's and RCCL's banners, warnings of the runtime -- goes to stderr instead: fd 1 is
    kept aside for the line and replaced by a copy of fd 2.  (Under torch.distributed.run the ranks share the
    launcher's stdout, so every rank does this, not only rank 0.)"""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)
    return _LINE_FD


def emit_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    fd = claim_stdout()
    while data:
        data = data[os.write(fd, data):]


def selftest_launch(args, world, rank):
    """the launcher alone (CPU, gloo): rendezvous, one collective, rank 0's line"""
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    ok = int(t.item()) == world * (world + 1) // 2 and dist.get_world_size() == world
    dist.barrier()
    dist.destroy_process_group()
    if args.selftest_launch == "fail" and rank == world - 1:
        return 3
    if rank == 0:
        emit_line({"selftest": "launch", "n_gpus": world, "sum_ok": bool(ok)})
    return 0 if ok else 4


# ----------------------------------------------------------------------------- helpers
def prewarm_clocks(ms):
    """Keep the GPU busy for `ms` milliseconds (fp32 matrix products, nothing of the workload) right before the warm-up
    steps.  Power management drops the clocks while the host prepares the case; they come back over the first ~15
    steps of 1 ms (k_rk 0.89 -> 0.80 ms, tools/dev/step_trend.py), which a warm-up of 5 steps does not cover and a
    run of thousands of steps does not see; 20 ms of products bring them back as well (a memory-bound load of the same
    length does not).  `config.clock_prewarm_ms` says what was done; --prewarm-ms 0 turns it off."""
    if ms <= 0:
        return
    import torch
    a = torch.randn(2048, 2048, device="cuda")
    b = torch.randn(2048, 2048, device="cuda")
    a @ b                       # (the first product loads the BLAS library: not counted)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            a @ b
        torch.cuda.synchronize()


def source_hash_of_profile(tag):
    p = os.path.join(ROOT, "profiles", tag, "pmc.json")
    try:
        d = json.load(open(p))
    except (OSError, ValueError):
        return None, None
    return d, d.get("_meta", {}).get("source_hash")


def pmc_traffic(tag, kernels, lib_hash):
    """HBM-side bytes per launch of the timed kernel(s) from the committed rocprofv3 PMC summary
    of THIS command (profiles/<tag>/pmc.json, separate --pmc passes, see profiles/tools/).
    Counters cannot be collected inside an ordinary run, so the number is the one measured for
    the default workload -- and only while the library is built from the very kernel sources the
    profile was taken with (pmc.json records their hash, the library exports its own); otherwise null."""
    d, prof_hash = source_hash_of_profile(tag)
    if d is None or prof_hash is None or prof_hash != lib_hash:
        return None, None, None
    tot = 0.0
    valu = {}
    for kern in kernels:
        hit = [v for k, v in d.items() if ("::" + kern + "(" in k or "::" + kern + "<" in k)]
        if not hit or "fetch_bytes_per_launch" not in hit[0] or "write_bytes_per_launch" not in hit[0]:
            return None, None, None
        tot += hit[0]["fetch_bytes_per_launch"] + hit[0]["write_bytes_per_launch"]
        g = lambda c: hit[0].get(c, {}).get("mean_per_launch")  # noqa: E731
        if g("SQ_WAVES") and g("SQ_WAVE_CYCLES"):
            valu[kern] = {"valu_instr_per_particle_step": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
                          "wave_cycles_valu_active": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
                          "wave_cycles_waiting": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")}
    return tot, "profiles/%s/pmc.json" % tag, valu


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


def algorithmic_bytes(slab, turb, ws):
    A = A_ADV_F32 if slab == "f32" else A_ADV_F64
    if turb:
        # + Kh 4 nodes x ws x 3 slots, + the materialised stream (124 draws
        # of 4 B written once and read once)
        A += (48 if slab == "f32" else 96) * ws + 124 * 4 * 2
    return A


VT_KERNELS = ["k_vt_table", "k_vt_sigs", "k_vt_norm", "k_vt_walk_fast", "k_vt_walk"]


def roofline_block(slab, turb, ws, n, k_avg_s, step_s, traffic, traffic_src, valu):
    """Bytes and time of one figure cover the SAME kernels.  RK4 only: the timed kernel is k_rk, which touches all of
    A_adv.  With turbulence the timed kernels are the VTurb group, whose algorithmic bytes are A_turb - A_adv (the Kh
    columns and the materialised stream); the whole step -- every kernel, launch gaps included -- is A_turb over the
    wall time of a step (`whole_step`)."""
    A_all = algorithmic_bytes(slab, turb, ws)
    A_k = A_all - algorithmic_bytes(slab, False, ws) if turb else A_all
    kernels = VT_KERNELS if turb else ["k_rk"]
    ach = (A_k * n / k_avg_s) / 1e9 if k_avg_s > 0 else 0.0
    ach_step = (A_all * n / step_s) / 1e9 if step_s > 0 else 0.0
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_algorithmic": (traffic / (A_k * n)) if traffic else None,
            "kernel": "+".join(kernels), "kernel_ms_avg": k_avg_s * 1e3,
            "algorithmic_bytes_per_launch": A_k * n, "algorithmic_bytes_per_particle_step": A_k,
            "valu_pmc": valu, "valu_issue": valu_issue(valu, kernels, n, k_avg_s),
            "whole_step": {"algorithmic_bytes_per_particle_step": A_all, "ms": step_s * 1e3, "achieved": ach_step,
                           "frac": ach_step / HBM_PEAK_GBS}}


ISLANDS_B = [(200, 215, 100, 140), (600, 640, 300, 310), (820, 830, 80, 90), (400, 402, 400, 460)]


def build_world(vi, uj, us, reflect):
    from ltrans_amd import roms_synth as rs
    g = rs.make_grid(vi, uj, us, islands=[(int(a * vi / 1024), int(b * vi / 1024), int(c * uj / 512), int(d * uj / 512))
                                          for a, b, c, d in ISLANDS_B] if reflect else ())
    geo = rs.build_geometry(g)
    bounds = rs.create_bounds(g)
    return g, geo, bounds


def make_parts(g, geo, bounds, numpar, seed, reflect):
    from ltrans_amd import roms_synth as rs
    if reflect:   # configs[4]: a fifth of the particles within 120 m of the west / south coast
        return rs.make_particles(g, geo, bounds, numpar, seed=seed, near_land=0.2, coast_band=(5.0, 120.0))
    return rs.make_particles(g, geo, bounds, numpar, seed=seed)


def behaviour_cfg(behavior, settle):
    """extra lt_config fields of the optional behaviour / settlement workload"""
    kw = dict(Behavior=behavior)
    if behavior:
        kw.update(swimstart=0.0, swimslow=0.002, swimfast=0.004, sink=-0.0003 if behavior == 6 else -0.002,
                  pediage=7200.0, deadage=90000.0, Sgradient=0.3, Hswimspeed=0.3)
    if settle:
        kw.update(settlementon=1, holesExist=1, mortality=1, pediage=3600.0, deadage=36000.0)
    return kw


def make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records, engine_cls=None):
    from ltrans_amd import roms_synth as rs
    from ltrans_amd.driver import LtransRun
    e = (engine_cls or ltrans_amd.Engine)(cfg)
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


class Stepper:
    """the internal-step loop of one engine: rotation outside the timed region"""

    def __init__(self, e, run, cfg):
        self.e, self.run, self.cfg = e, run, cfg
        self.p = 2                      # second external step: plain (b,c,f) time interpolation
        self.ex = run.external_times(self.p)
        self.stepIT = cfg.dt // cfg.idt
        self.it = 0

    def needs_rotation(self):
        return self.it + 1 > self.stepIT

    def rotate(self):
        """next external step: updateHydro = rotate slots + upload ONE record.  Kept
        outside the timed region (slab upload is reported separately)."""
        self.p += 1
        self.it = 0
        self.run.p = self.p
        self.run.update_hydro()
        self.ex = self.run.external_times(self.p)

    def next_times(self):
        self.it += 1
        return self.run.internal_times(self.ex, self.it)


def timed_steps(st, nsteps, one_step, fence):
    wall = upload = 0.0
    t0 = time.perf_counter()
    for _ in range(nsteps):
        if st.needs_rotation():
            fence()
            wall += time.perf_counter() - t0
            tu = time.perf_counter()
            st.rotate()
            fence()
            upload += time.perf_counter() - tu
            t0 = time.perf_counter()
        one_step()
    fence()
    wall += time.perf_counter() - t0
    return wall, upload


def sustained_series(e, run, st, one_step, fence, nsteps, make_rom, dims):
    """`nsteps` steps of the engine as a host runs it for hours: nothing excluded.  The re-sorts of the particle order
    (one in 32 steps, inside lt_step) and the slab rotations are inside the timed series, the rotations through the
    pipeline (lt_slab_prefetch of the next record right after each lt_slab_commit, float32 host arrays as NetCDF
    delivers a ROMS history file); only the synthesis of the records -- the stand-in for reading the file -- is done
    before the clock starts (and the first record is handed to the pipeline there, which allocates its staging
    buffers on first use).  -> (wall seconds, rotations, re-sorts)"""
    stepIT = st.stepIT
    n_rot = (st.it + nsteps - 1) // stepIT if st.it + nsteps > stepIT else 0
    first = run.stepf                              # 0-based history record the next rotation installs
    roms = {r: make_rom(r) for r in range(first, first + n_rot)}
    run.rom_fn, run.dims, run._prefetched = (lambda r: roms[r]), dims, None
    left_rot = n_rot
    if left_rot:
        # the record the first rotation installs is handed over before the clock starts: the first lt_slab_prefetch of an
        # engine allocates the library's pinned staging buffers (~27 ms for this grid, once per engine)
        run.prefetch_next()
    fence()
    resorts0 = e.counters()["resorts"]
    t0 = time.perf_counter()
    trend = int(os.environ.get("LT_BENCH_TREND", "0"))   # (diagnosis: a fence and a line on stderr every so many steps)
    tb = t0
    for i in range(nsteps):
        if st.needs_rotation():
            st.rotate()                            # lt_slab_commit: the prefetched record becomes slot t_f
            left_rot -= 1
            if left_rot > 0:
                run.prefetch_next()                # the next record's upload runs beside this external step's steps
        one_step()
        if trend and (i + 1) % trend == 0:
            fence()
            sys.stderr.write("sustained steps %d..%d: %.3f ms per step\n" % (i + 2 - trend, i + 1, (time.perf_counter() - tb) * 1e3 / trend))
            tb = time.perf_counter()
    fence()
    wall = time.perf_counter() - t0
    run.rom_fn, run.dims, run._prefetched = None, None, None
    return wall, n_rot - left_rot, e.counters()["resorts"] - resorts0


def hydro_update_costs(e, run, rs, g, geo, turb, st, one_step, fence):
    """What one updateHydro costs on this box, outside the throughput measurement
    (SURVEY.md 8f row 1): the synchronous lt_set_slab path (host waits, GPU idle) against
    the slab pipeline (lt_slab_prefetch beside the steps, lt_slab_commit at the rotation).
    ``exposed`` = wall of [prefetch, stepIT steps, commit] - wall of [stepIT steps]."""
    import numpy as np
    stepIT = run.dt // run.idt
    r0 = run.stepf
    masked = rs.make_record(g, geo, r0, with_kh=bool(turb))
    rom32 = rs.make_record(g, geo, r0 + 1, with_kh=bool(turb), raw=np.float32)
    rom64 = rs.make_record(g, geo, r0 + 2, with_kh=bool(turb), raw=np.float64)
    dims = (g["vi"], g["uj"], g["ui"], g["vj"])
    mb = lambda d: sum(v.nbytes for v in d.values() if v is not None) / 1e6  # noqa: E731

    def steps():
        for _ in range(stepIT):
            st.it = st.it % stepIT
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


def cpu_baseline(ltrans_amd, args, spec, g, geo, bounds, records):
    """Time the CPU oracle (kind "port": single-thread C restatement of the
    reference path, oracle/ltrans_oracle.c) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    n = args.cpu_sample
    parts = make_parts(g, geo, bounds, n, 777, spec["reflect"])
    turb = 1 if spec["turb"] else 0
    cfg = ltrans_amd.LtConfig(numpar=n, us=spec["us"], ws=spec["us"] + 1, dt=3600, idt=120,
                              ErrorFlag=1, HTurbOn=turb, VTurbOn=turb, **behaviour_cfg(spec["behavior"], spec["settle"]))
    e, run = make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records, engine_cls=oracle_lib.OracleEngine)
    ex = run.external_times(2)
    t0 = time.perf_counter()
    for it in range(1, args.cpu_steps + 1):
        e.update_particles(ex, run.internal_times(ex, it), 2, it)
    dt = time.perf_counter() - t0
    e.close()
    # what the port costs against the reference's own Fortran where that compiles (oracle/_ref: the leaf modules,
    # flang -O2): the same inputs through the same loops, oracle/time_leaves.py; None when _ref is not there
    leaf = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import time_leaves
        leaf = time_leaves.measure(0.3)
    except Exception as ex:          # noqa: BLE001 -- an extra of the baseline, never a reason to lose the line
        sys.stderr.write(f"bench.py: leaf timing skipped: {ex}\n")
    return {"value": n * args.cpu_steps / dt, "unit": "particle-steps/s", "cores": 1,
            "kind": "port",
            "port_vs_reference_leaf_ratio": (leaf["advection_step"]["ratio"] if leaf else None),
            "port_vs_reference_leaves": leaf,
            "sample": f"{n} particles x {args.cpu_steps} internal steps on the same "
                      f"{spec['grid']} grid, {dt:.1f} s wall, {os.cpu_count()} host cores present"}


def resolve_workload(args, world, name=None):
    """--workload auto: the configuration the metric is quoted on (configs[1]; configs[2] with --turb) PER GPU, i.e.
    weak scaling over the ranks.  An explicit --workload is taken whole (strong scaling unless --weak).
    name="config4": the embedded configs[3] line of an auto run (its own --configs3-grid / --configs3-numpar)."""
    embedded = name is not None
    auto = not embedded and args.workload == "auto"
    if name is None:
        name = args.workload
    if name == "auto":
        name = "config3" if args.turb else "config2"
    grid, total, turb, reflect = WORKLOADS[name]
    if args.turb and not embedded:
        turb = True
    grid = (args.configs3_grid if embedded else args.grid) or grid
    vi, uj, us = (int(v) for v in grid.split("x"))
    behavior = args.behavior if (args.behavior is not None and not embedded) else (6 if reflect else 0)
    settle = bool((args.settle and not embedded) or reflect)
    weak = bool(world > 1 and not embedded and (args.weak or auto))
    if weak:
        per_rank = args.numpar or 1_000_000
        total = per_rank * world
    else:
        total = (args.configs3_numpar if embedded else args.numpar) or total
    return dict(name=name, grid=grid, vi=vi, uj=uj, us=us, turb=turb, reflect=reflect, behavior=behavior,
                settle=settle, total=total, weak=weak)


def workload_text(spec, world, n_local, backend="nccl"):
    s = (f"{spec['total']} particles in total" if not spec["weak"] else f"{n_local} particles per GPU")
    s += f", synthetic {spec['grid']} ROMS grid, "
    s += ("RK4 + HTurb + VTurb (MT19937 stream on device), idt=120 s" if spec["turb"]
          else "RK4 advection only (HTurb/VTurb off), idt=120 s")
    s += (f", behaviour {spec['behavior']}" if spec["behavior"] else ", behavior off")
    s += (", islands + drift into the coast (reflection), settlement + mortality" if spec["reflect"]
          else (", settlement + mortality" if spec["settle"] else ""))
    if world > 1:
        coll = "RCCL" if backend == "nccl" else f"{backend} (host memory: a rehearsal, not RCCL)"
        s += f", sharded over {world} GPUs by contiguous index block, {coll} all-gather of (x,y,z) per step"
    return s


# ----------------------------------------------------------------------------- main
def main():
    t_start = time.perf_counter()
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_children(args)                     # before anything touches torch.cuda / HIP
    claim_stdout()                                       # before any library can write to it
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values\n")
        return 2
    if args.selftest_launch:
        return selftest_launch(args, world, rank)

    import numpy as np
    import torch
    import ltrans_amd
    from ltrans_amd import roms_synth as rs
    from ltrans_amd import sharding

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
        assert dist.get_world_size() == args.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    def measure(spec, main_line):
        """One workload on all ranks -> (rc, the JSON object of the line as rank 0 sees it).  main_line: with the
        one-GPU extras (configs[2] timed beside, updateHydro / print-step / init-geometry costs, the CPU baseline)."""
        t_begin = time.perf_counter()
        vi, uj, us = spec["vi"], spec["uj"], spec["us"]
        # (--emulate-rank: the shard and the stream window of rank R of W, computed by this one process)
        em_rank, em_world = (int(x) for x in args.emulate_rank.split(":")) if args.emulate_rank else (rank, world)
        if args.emulate_rank and (world != 1 or not 0 <= em_rank < em_world):
            sys.stderr.write("bench.py: --emulate-rank R:W needs --gpus 1 and 0 <= R < W\n")
            return 2, None
        lo, hi = sharding.shard_range(spec["total"], em_world, em_rank)
        n = hi - lo
        g, geo, bounds = build_world(vi, uj, us, spec["reflect"])
        drift = (-0.2, -0.15) if spec["reflect"] else (0.0, 0.0)
        # every rank's shard is its own draw (seed + rank): the shards of a strong-scaling run are the
        # blocks of the global particle list that rank 0 re-assembles for the single-GPU reference
        parts = make_parts(g, geo, bounds, n, 12345 + em_rank, spec["reflect"])
        rec_cache = {}
        needs_st = spec["behavior"] not in (0, 6)     # salinity / temperature only for the behaviours that read them

        def records(r, with_kh=None):
            k = (r, bool(spec["turb"] if with_kh is None else with_kh))
            if k not in rec_cache:
                rec_cache[k] = rs.make_record(g, geo, r, with_kh=k[1], drift=drift, with_st=needs_st)
            return rec_cache[k]

        def config_for(numpar, turb):
            t = 1 if turb else 0
            return ltrans_amd.LtConfig(numpar=numpar, us=us, ws=us + 1, dt=3600, idt=120, ErrorFlag=1, device=dev.index,
                                       HTurbOn=t, VTurbOn=t, ConstantHTurb=1.0, **behaviour_cfg(spec["behavior"], spec["settle"]),
                                       slab_storage=(ltrans_amd.abi.LT_SLAB_F32 if args.slab == "f32"
                                                     else ltrans_amd.abi.LT_SLAB_F64))

        cfg = config_for(n, spec["turb"])
        e, run = make_engine(ltrans_amd, cfg, g, geo, bounds, parts, records)
        st = Stepper(e, run, cfg)
        lib_hash = e.source_hash()

        # N > 1: the end-of-step all-gather of (x,y,z) runs on its own stream and overlaps the next
        # step's kernels.  It is ordered behind the step on the DEVICE (an event on the engine's stream;
        # no host synchronisation per step); the packed positions are double-buffered and the engine's
        # stream waits for the all-gather that last read a buffer before packing into it again.
        nmax = max(sharding.shard_sizes(spec["total"], world)) if world > 1 else n
        send = recv = gather_stream = eng_stream = None
        send_done = [None, None]
        nstep = [0]
        if world > 1:
            send = [torch.zeros(3 * nmax, dtype=torch.float64, device=dev) for _ in range(2)]
            recv = torch.empty(3 * nmax * world, dtype=torch.float64, device=dev)
            gather_stream = torch.cuda.Stream(device=dev)
            eng_stream = torch.cuda.ExternalStream(e.stream(), device=dev)

        draws = spec["turb"] or 1 <= spec["behavior"] <= 5
        # A shard's window of the single serial random stream is static while no particle can stop drawing:
        # nothing is unreleased (DOB <= 0), settles, dies or leaves through an open boundary.  Otherwise the
        # ranks exchange their counts before every step (one all-gather of one int64 each, one host read).
        # The choice is the JOB's, not the rank's (a rank that exchanged counts while another did not would wait for
        # ever): the minimum over the ranks of the local predicate.  The engine checks it on the device in every step
        # (lt_set_rng_window_static: a step whose draw total differs makes e.counters() fail below).
        static_window = (draws and not (1 <= spec["behavior"] <= 5) and not spec["settle"]
                         and float(parts["par"][:, 10].max()) <= 0.0)
        if world > 1:
            sw = torch.tensor([1 if static_window else 0], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(sw, op=dist.ReduceOp.MIN)
            static_window = bool(int(sw.item()))
        if static_window:
            # (one engine alone: its window is the whole stream -- set once, no step waits for the device)
            sizes = sharding.shard_sizes(spec["total"], em_world)
            e.set_rng_window(int(sum(sizes[:em_rank])), int(sum(sizes)), persistent=True)
        elif em_world > world and draws:
            sys.stderr.write("bench.py: --emulate-rank needs a workload with a static stream window\n")
            return 2, None

        # The per-step exchange of the shards' draw counts: inside the library (lt_exchange_rng_window: ncclAllGather of
        # one int64 per rank on the engine's own RCCL communicator, whose id travels over torch.distributed once) where
        # every rank has a GPU of its own; through torch.distributed (sharding.rng_window: one host read per step) in
        # the gloo rehearsals, where the ranks share a card and RCCL refuses a second rank on a device.
        abi_exchange = False
        if world > 1 and draws and not static_window and backend == "nccl" and not os.environ.get("LT_BENCH_TORCH_EXCHANGE"):
            try:
                box = [e.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                e.comm_init(box[0], world, rank)
                abi_exchange = True
            except Exception as ex:      # noqa: BLE001 -- the torch path below does the same job
                sys.stderr.write(f"bench.py: rank {rank}: lt_comm_init failed ({ex}); exchanging through torch.distributed\n")
            ok_t = torch.tensor([1 if abi_exchange else 0], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)      # (the choice is the job's: every rank or none)
            abi_exchange = bool(int(ok_t.item()))

        def one_step():
            ix = st.next_times()
            if abi_exchange:
                e.exchange_rng_window(st.ex, ix, st.p, st.it)
            elif world > 1 and draws and not static_window:
                if 1 <= spec["behavior"] <= 5:   # data-dependent number of deviates: count draws
                    mine = e.count_draws(st.ex, ix, st.p, st.it)
                    before, total = sharding.rng_window(dist, torch, mine, world, rank, device=coll_dev)
                    e.set_rng_window_draws(before, total)
                else:
                    before, total = sharding.rng_window(dist, torch, e.count_active(ix), world, rank, device=coll_dev)
                    e.set_rng_window(before, total)
            e.update_particles(st.ex, ix, st.p, st.it)
            if world > 1:
                b = nstep[0] & 1
                nstep[0] += 1
                if send_done[b] is not None:
                    eng_stream.wait_event(send_done[b])   # the all-gather that last read this buffer
                e.pack_positions(send[b].data_ptr())
                packed = torch.cuda.Event()
                packed.record(eng_stream)
                gather_stream.wait_event(packed)
                with torch.cuda.stream(gather_stream):
                    if backend == "nccl":
                        dist.all_gather_into_tensor(recv, send[b])
                    else:                                   # rehearsal: the same ordering, the collective through host memory
                        parts_ = [torch.empty(3 * nmax, dtype=torch.float64) for _ in range(world)]
                        dist.all_gather(parts_, send[b].cpu())      # (.cpu() on the gather stream: behind `packed`)
                        recv.copy_(torch.cat(parts_), non_blocking=True)
                    send_done[b] = torch.cuda.Event()
                    send_done[b].record(gather_stream)

        def fence():
            e.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        setup_s = time.perf_counter() - t_begin
        prewarm_clocks(args.prewarm_ms)
        for _ in range(args.warmup):
            if st.needs_rotation():
                st.rotate()
            one_step()
        fence()
        resorts0 = e.counters()["resorts"]
        e.profile_begin(args.steps)
        wall, upload_s = timed_steps(st, args.steps, one_step, fence)
        k_ms, k_n = e.profile_end()
        # (a rank that fails here must not leave the others waiting in the collectives below: the verdict is the job's)
        rc_local, stats = 0, None
        try:
            stats = e.counters()      # (fails if a static stream window did not hold)
            if static_window and stats["active"] != n:
                sys.stderr.write(f"bench.py: rank {rank}: static stream window, but {stats['active']} of {n} particles were tracked\n")
                rc_local = 3
        except Exception as ex:       # noqa: BLE001 -- reported, then every rank leaves together
            sys.stderr.write(f"bench.py: rank {rank}: {ex}\n")
            rc_local = 3
        rc_t = torch.tensor([rc_local], dtype=torch.int64, device=coll_dev)
        if world > 1:
            dist.all_reduce(rc_t, op=dist.ReduceOp.MAX)
        if int(rc_t.item()) != 0:
            return int(rc_t.item()), None
        # what a device-side re-sort of the particle order costs (one every 32 steps inside lt_step): if none has
        # happened yet, a few untimed steps more until one has
        resort_ms = stats["last_resort_ms"] if stats["resorts"] else None
        if resort_ms is None and not args.emulate_rank:
            for _ in range(40):
                if st.needs_rotation():
                    st.rotate()
                one_step()
                c2 = e.counters()
                if c2["resorts"]:
                    resort_ms = c2["last_resort_ms"]
                    break
            fence()
        resorts_in = stats["resorts"] - resorts0
        resort_excl = (max(args.steps / 32.0 - resorts_in, 0.0) * resort_ms * 1e-3) if resort_ms is not None else None

        wall_t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        wall = float(wall_t.item())

        # (an emulated rank reports its own shard's particle-steps: one GPU's share of the job)
        total_particle_steps = float(n if args.emulate_rank else spec["total"]) * args.steps
        value = total_particle_steps / wall
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        kernels = VT_KERNELS if spec["turb"] else ["k_rk"]
        default_shape = (n == 1_000_000 and spec["grid"] == "512x256x30" and args.slab == "f32" and not spec["reflect"])
        traffic = traffic_src = valu = None
        if default_shape:
            traffic, traffic_src, valu = pmc_traffic(PROFILE_TURB if spec["turb"] else PROFILE_ADV, kernels, lib_hash)

        out = {
            "metric": "particle-steps/s", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if (spec["weak"] or world == 1) else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text(spec, world, n, backend) + (f"; THIS LINE: rank {em_rank} of {em_world} emulated by one "
                       "process (its shard, its window of the random stream, no all-gather)" if args.emulate_rank else ""),
                       "baseline_config": spec["name"],
                       "particles_total": spec["total"], "particles_this_rank": n, "grid": spec["grid"],
                       "slab_storage": args.slab, "kernel_source_hash": lib_hash,
                       "active_last_step": stats["active"], "errors_last_step": stats["errors"],
                       "sigerr_last_step": stats["sigerr"], "slab_rotation_s_excluded": upload_s,
                       # boundary queries of the last step that left the segment index (a move across more than two of
                       # its cells: all segments are tested) / point-in-polygon tests that needed the reference's full
                       # algorithm (a vertex on the ray), VTurb intervals iterated, particle-steps redone by k_rk_slow
                       "brute_bounds_last_step": stats["brute_bounds"], "pip_full_last_step": stats["pip_full"],
                       "vt_queued_last_step": stats["vt_queued"], "rk_slow_last_step": stats["rk_slow"],
                       # the device-side re-sort of the particle order (every `resort_every` = 32 steps inside lt_step):
                       # how many fell into the timed region -- they are part of `value` -- and what one costs
                       "clock_prewarm_ms": args.prewarm_ms,
                       "resorts_in_timed_region": resorts_in, "resort_every_steps": 32, "resort_ms_each": resort_ms,
                       # the re-sorts a run of `steps` steps has on average (steps / 32) that did not fall into the
                       # timed region, in seconds: add it to the region's wall time for the steady-state figure
                       "resort_s_excluded": resort_excl,
                       "rng_window": (("static (validated on the device every step)" if static_window
                                       else ("exchanged per step by lt_exchange_rng_window (RCCL inside the library)" if abi_exchange
                                             else "exchanged per step")) if draws else None),
                       "collective_backend": (backend if world > 1 else None), "device_index": dev_index,
                       "setup_s": setup_s},
            "roofline": roofline_block(args.slab, spec["turb"], us + 1, n, k_avg_s, wall / args.steps, traffic, traffic_src,
                                       valu),
        }

        # flat copies of the figures a reader of the record needs (a record that keeps only scalar keys keeps these)
        rf = out["roofline"]
        out["config"].update({
            "kernel_ms_avg": rf["kernel_ms_avg"], "roofline_frac": rf["frac"], "whole_step_frac": rf["whole_step"]["frac"],
            "valu_issue_frac": (rf["valu_issue"] or {}).get("frac"),
            "traffic_over_algorithmic": rf["traffic_over_algorithmic"],
            # k_rk's workgroups go to the XCDs in chunks of this many (lt_rk_launch.h): an XCD's L2 then serves eight
            # patches of the domain instead of one contiguous eighth -- +0.17 GB of HBM reads per launch (1.17x the
            # algorithmic bytes instead of 0.95x) for an even load late in a run
            "rk_xcd_chunk": 64})

        if main_line and world == 1 and args.sustained_steps > 0 and not args.emulate_rank:
            dims = (g["vi"], g["uj"], g["ui"], g["vj"])
            w_s, rot_s, res_s = sustained_series(
                e, run, st, one_step, fence, args.sustained_steps,
                lambda r: rs.make_record(g, geo, r, with_kh=bool(spec["turb"]), drift=drift, with_st=needs_st, raw=np.float32), dims)
            out["config"].update({
                "sustained_steps": args.sustained_steps, "sustained_value": n * args.sustained_steps / w_s,
                "sustained_ms_per_step": w_s / args.sustained_steps * 1e3,
                "sustained_resorts_inside": res_s, "sustained_slab_rotations_inside": rot_s,
                "sustained_note": "the same engine, right after the timed window: re-sorts and slab rotations (lt_slab_prefetch / "
                                  "lt_slab_commit, float32 host arrays) inside the series; no clock pre-warm of its own"})

        if main_line and world == 1 and spec["name"] == "config2" and not args.no_turbulence_line:
            # BASELINE configs[2] in the same line: the same particles with HTurb + VTurb, same number of steps
            cfg_t = config_for(n, True)
            t_create = time.perf_counter()
            e_t, run_t = make_engine(ltrans_amd, cfg_t, g, geo, bounds, parts, lambda r: records(r, True))
            st_t = Stepper(e_t, run_t, cfg_t)
            static_t = float(parts["par"][:, 10].max()) <= 0.0 and not spec["settle"]
            if static_t:
                e_t.set_rng_window(0, n, persistent=True)

            def step_t():
                e_t.update_particles(st_t.ex, st_t.next_times(), st_t.p, st_t.it)

            def fence_t():
                e_t.sync()
                torch.cuda.synchronize()

            step_t()                  # the first step carries the start-up of the random stream (generator states, ring)
            fence_t()
            first_step_s = time.perf_counter() - t_create
            prewarm_clocks(args.prewarm_ms)
            for _ in range(max(args.warmup - 1, 0)):
                if st_t.needs_rotation():
                    st_t.rotate()
                step_t()
            fence_t()
            rs0 = e_t.counters()["resorts"]
            e_t.profile_begin(args.steps)
            wall_tb, _ = timed_steps(st_t, args.steps, step_t, fence_t)
            kt_ms, kt_n = e_t.profile_end()
            kt_avg = (kt_ms / max(kt_n, 1)) * 1e-3
            tr_t, src_t, valu_t = pmc_traffic(PROFILE_TURB, VT_KERNELS, lib_hash) if default_shape else (None, None, None)
            cnt_t = e_t.counters()
            out["config"]["turbulence"] = {
                "workload": "BASELINE configs[2]: the same + HTurb + VTurb (MT19937 stream on device)",
                "value": n * args.steps / wall_tb, "unit": "particle-steps/s", "ms_per_step": wall_tb / args.steps * 1e3,
                "steps": args.steps, "sigerr_last_step": cnt_t["sigerr"],
                "rng_window": "static (validated on the device every step)" if static_t else "the engine's own draws, read back per step",
                "resorts_in_timed_region": cnt_t["resorts"] - rs0,
                "create_to_first_step_s": first_step_s,
                "roofline": roofline_block(args.slab, True, us + 1, n, kt_avg, wall_tb / args.steps, tr_t, src_t, valu_t)}
            rft = out["config"]["turbulence"]["roofline"]
            out["config"].update({
                "turb_value": n * args.steps / wall_tb, "turb_ms_per_step": wall_tb / args.steps * 1e3,
                "turb_vt_kernels_ms": rft["kernel_ms_avg"], "turb_roofline_frac": rft["frac"],
                "turb_whole_step_frac": rft["whole_step"]["frac"],
                "turb_valu_issue_frac": (rft["valu_issue"] or {}).get("frac"),
                "turb_traffic_over_algorithmic": rft["traffic_over_algorithmic"]})
            if args.sustained_steps > 0:
                dims_t = (g["vi"], g["uj"], g["ui"], g["vj"])
                ns_t = max(args.sustained_steps // 4, 1)         # (a turbulence step is 4-5x an advection step)
                w_t, rot_t, res_t = sustained_series(
                    e_t, run_t, st_t, step_t, fence_t, ns_t,
                    lambda r: rs.make_record(g, geo, r, with_kh=True, drift=drift, with_st=needs_st, raw=np.float32), dims_t)
                out["config"].update({"turb_sustained_steps": ns_t, "turb_sustained_value": n * ns_t / w_t,
                                      "turb_sustained_ms_per_step": w_t / ns_t * 1e3,
                                      "turb_sustained_resorts_inside": res_t, "turb_sustained_slab_rotations_inside": rot_t})
            e_t.close()

        if main_line and world == 1 and not args.no_hydro_timing:
            out["config"]["hydro_update"] = hydro_update_costs(e, run, rs, g, geo, spec["turb"], st, one_step, fence)
            # the print step: lon/lat/status on the device + download of the
            # writers' columns, against downloading the whole particle state
            fence()
            e.get_output()
            t0 = time.perf_counter(); e.get_output(); t_out = time.perf_counter() - t0
            # (the first download allocates the library's pinned staging buffer and the host's arrays; a host that
            # prints every few steps keeps both)
            held = e.get_particles()
            t0 = time.perf_counter(); held = e.get_particles(out=held); t_all = time.perf_counter() - t0
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
        e.close()

        if world > 1 and not spec["weak"] and not args.no_single_ref:
            # strong scaling needs its one-GPU point on the SAME workload: rank 0 alone runs all the
            # particles (the concatenation of the ranks' shards) for a few steps; reported raw, beside `value`.
            # It costs rank 0 the generation of every shard's particles and a whole-workload engine: skipped (and
            # said so) when the run has already used more than --single-ref-budget seconds; --no-single-ref skips it.
            # (the other ranks hand their shards over through files in the node's temporary directory: generating
            # them again on rank 0 took longer than everything else together)
            # (a directory rank 0 makes for this run -- mkdtemp: unpredictable name, owner-only -- and removes again)
            import shutil
            import tempfile
            box = [tempfile.mkdtemp(prefix="ltbench_") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            shard_dir = box[0]
            seen = torch.tensor([1 if os.path.isdir(shard_dir) else 0], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(seen, op=dist.ReduceOp.MIN)
            same_node = int(seen.item()) == 1      # (ranks on another node do not see rank 0's directory)
            if rank > 0 and same_node:
                np.savez(os.path.join(shard_dir, f"rank{rank}.npz"), **parts)
            torch.cuda.synchronize()
            dist.barrier()
            elapsed = time.perf_counter() - t_start
            if rank == 0 and not same_node:
                out["config"]["single_gpu_same_workload"] = {"skipped": "the ranks do not share a node"}
            elif rank == 0 and elapsed > args.single_ref_budget:
                out["config"]["single_gpu_same_workload"] = {"skipped": f"{elapsed:.0f} s used, budget {args.single_ref_budget} s"}
            elif rank == 0:
              try:
                t_ref = time.perf_counter()
                shards = [parts]
                for r in range(1, world):
                    with np.load(os.path.join(shard_dir, f"rank{r}.npz")) as z:
                        shards.append({k: z[k] for k in parts})
                allp = {k: np.concatenate([s[k] for s in shards]) if shards[0][k].ndim == 1
                        else np.asfortranarray(np.vstack([s[k] for s in shards])) for k in parts}
                del shards
                cfg1 = config_for(spec["total"], spec["turb"])
                e1, run1 = make_engine(ltrans_amd, cfg1, g, geo, bounds, allp, records)
                st1 = Stepper(e1, run1, cfg1)
                k1 = max(2, min(args.steps, 5))

                def step1():
                    e1.update_particles(st1.ex, st1.next_times(), st1.p, st1.it)

                def fence1():
                    e1.sync()

                step1(); step1(); fence1()
                w1, _ = timed_steps(st1, k1, step1, fence1)
                out["config"]["single_gpu_same_workload"] = {
                    "value": spec["total"] * k1 / w1, "unit": "particle-steps/s", "ms_per_step": w1 / k1 * 1e3, "steps": k1,
                    "note": "all the particles on rank 0's GPU alone, no all-gather",
                    "wall_s_incl_setup": time.perf_counter() - t_ref}
                e1.close()
              except Exception as ex:    # noqa: BLE001 -- the reference point is an extra: the line goes out without it
                sys.stderr.write(f"bench.py: single-GPU reference failed: {ex}\n")
                out["config"]["single_gpu_same_workload"] = {"skipped": f"failed: {ex}"}
              finally:
                shutil.rmtree(shard_dir, ignore_errors=True)
            dist.barrier()
            if rank == 0:
                shutil.rmtree(shard_dir, ignore_errors=True)

        if main_line and rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ltrans_amd, args, spec, g, geo, bounds, records)
        elif rank == 0:
            out["cpu_baseline"] = None
        return 0, out

    spec = resolve_workload(args, world)
    rc, out = measure(spec, True)
    if rc:
        return rc
    if world > 1 and args.workload == "auto" and not args.weak and not args.no_configs3_line:
        # BASELINE configs[3] -- the 8-GPU configuration by name: 10M particles in total, 1024x512x50, HTurb + VTurb --
        # strong-scaled over the same ranks, with its one-GPU point; reported raw under config.configs3_strong
        rc, o4 = measure(resolve_workload(args, world, name="config4"), False)
        if rc:
            return rc
        if rank == 0:
            c4 = o4["config"]
            out["config"]["configs3_strong"] = {
                "workload": c4["workload"], "value": o4["value"], "unit": o4["unit"], "ms_per_step": o4["ms_per_step"],
                "steps": o4["steps"], "scaling": o4["scaling"], "particles_total": c4["particles_total"],
                "particles_this_rank": c4["particles_this_rank"], "grid": c4["grid"], "rng_window": c4["rng_window"],
                "errors_last_step": c4["errors_last_step"], "sigerr_last_step": c4["sigerr_last_step"],
                "setup_s": c4["setup_s"], "roofline": o4["roofline"],
                "single_gpu_same_workload": c4.get("single_gpu_same_workload")}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["config"]["total_s"] = time.perf_counter() - t_start
        emit_line(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
