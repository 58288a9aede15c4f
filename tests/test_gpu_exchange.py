"""lt_exchange_rng_window: SURVEY 8e's second collective -- the shards' draw counts of the coming step, exchanged
inside the library (one int64 per rank over the engine's RCCL communicator) -- for the runs in which a particle can
stop or start drawing (settlement, mortality, late releases, behaviours with data-dependent draws; LTRANS.f90:790-816,
random_module.f90:166-197).  The box has one GPU and RCCL refuses two ranks on a device, so
  * the N-rank arithmetic (the window a shard takes from G gathered counts, the count it sends from the device) runs
    through the test build, which puts what the collective would deliver into the receive buffer;
  * the collective itself runs with one rank on the product library (ncclAllGather of one int64 on the communicator's
    stream, behind the event that says the prefix sum has run);
  * the Fortran harness runs a settlement + mortality + late-release case in rank mode."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
import hip_test_lib
import ltrans_amd
from ltrans_amd import casefile, sharding
from ltrans_amd.driver import LtransRun

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "ltransv.2b_amd", "fortran", "ltrans_hip_driver")
INTS = ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand", "dead", "oob", "settle", "endpoly")


class ExchangeEngine(hip_test_lib.HipTestEngine):
    def test_exchange(self, ex, ix, p, it, nranks, rank, counts):
        ex = np.ascontiguousarray(ex, dtype=np.float64)
        ix = np.ascontiguousarray(ix, dtype=np.float64)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        f = self.lib.lt_test_exchange_rng_window
        f.restype = C.c_int
        self._check(f(self._h, ex.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), int(p), int(it), int(nranks),
                      int(rank), counts.ctypes.data_as(C.c_void_p)), "test_exchange_rng_window")


def _dynamic_case(name="tiny_settle", **extra):
    """settlement + mortality + HTurb + VTurb, a fifth of the particles released during the run: the shards' draw
    counts differ and change from step to step"""
    kw = dict(HTurbOn=1, VTurbOn=1, ConstantHTurb=1.0, pediage=600.0, deadage=1200.0)   # settling from step 5, dying from step 10
    kw.update(extra)
    case = cases.build_case(name, **kw)
    n = case["cfg"].numpar
    dob = np.zeros(n)
    dob[3::5] = 120.0 * (1 + np.arange(len(dob[3::5])) % 9)     # released at internal steps 1 .. 9
    case["parts"]["par"][:, 10] = dob
    return case


def _shard(case, lo, hi):
    shard = dict(case)
    shard["cfg"] = case["cfg"].copy(numpar=hi - lo)
    shard["parts"] = {k: (v[lo:hi] if v.ndim == 1 else v[lo:hi, :]) for k, v in case["parts"].items()}
    return shard


def _full(case, steps):
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()
    run.run(1, max_internal=steps)
    out = e.get_particles()
    out["rng"] = e.get_rng()
    e.close()
    return out


@pytest.mark.parametrize("name,bounds_", [("tiny_settle", [0, 200, 200, 768]), ("tiny_settle", [0, 301, 550, 768]),
                                          ("tiny_beh4", [0, 100, 300, 512])])
def test_three_shards_through_the_exchange_reproduce_the_unsharded_run(name, bounds_):
    """three ragged shards (one of them empty in the first layout).  Set A does what a host with its own collective
    does (lt_count_draws, the host's sum, lt_set_rng_window_draws); set B gets A's counts of the same step as "what
    the collective delivers" and goes through lt_exchange_rng_window -- its own count comes from ITS device, the
    prefix sum runs inside the call.  Both equal the unsharded run bit for bit, integers included, and hand back the
    same generator state."""
    steps = 12
    case = _dynamic_case(name) if name == "tiny_settle" else cases.build_case(name)
    if name != "tiny_settle":
        case["parts"]["par"][3::5, 10] = 360.0
    full = _full(case, steps)
    world = len(bounds_) - 1
    sets = {}
    for tag, cls in (("A", ltrans_amd.Engine), ("B", ExchangeEngine)):
        engines = []
        for r in range(world):
            sh = _shard(case, bounds_[r], bounds_[r + 1])
            e = cls(sh["cfg"])
            e.set_grid(sh["geo"]); e.set_bounds(sh["bounds"])
            if sh["habitat"] is not None:
                e.set_habitat(sh["habitat"])
            e.set_particles(**sh["parts"])
            run = LtransRun(e, 3600, 120, case["record"])
            run.init_hydro()
            engines.append((e, run))
        sets[tag] = engines
    ex = sets["A"][0][1].external_times(1)
    changed = set()
    for it in range(1, steps + 1):
        ix = sets["A"][0][1].internal_times(ex, it)
        counts = [e.count_draws(ex, ix, 1, it) for e, _ in sets["A"]]
        changed.add(tuple(counts))
        for r, (e, _) in enumerate(sets["A"]):
            e.set_rng_window_draws(sum(counts[:r]), sum(counts))
            e.update_particles(ex, ix, 1, it)
        for r, (e, _) in enumerate(sets["B"]):
            given = list(counts)
            given[r] = -12345                    # (this rank's own entry is not taken from the host)
            e.test_exchange(ex, ix, 1, it, world, r, given)
            e.update_particles(ex, ix, 1, it)
            assert e.stats()["rng_draws"] == counts[r]
    assert len(changed) >= 4                     # the counts did change during the run
    for tag in ("A", "B"):
        got = [e.get_particles() for e, _ in sets[tag]]
        assert np.array_equal(np.vstack([g["par"] for g in got]), full["par"]), tag
        for k in INTS:
            assert np.array_equal(np.concatenate([g[k] for g in got]), full[k]), (tag, k)
        for e, _ in sets[tag]:
            mt, mti = e.get_rng()
            assert mti == full["rng"][1] and np.array_equal(mt, full["rng"][0]), tag
            e.close()
    assert full["settle"].sum() > 0 or name != "tiny_settle"


@pytest.mark.parametrize("how", ["file", "id"])
def test_exchange_on_a_one_rank_communicator(tmp_path, how):
    """the real collective (ncclAllGather of one int64 on the communicator's stream) with the one rank the box allows:
    exchange + step, every step, equals the plain run.  The communicator through the id file (the Fortran harness's
    way) and through lt_comm_unique_id / lt_comm_init (bench.py's way: the id travels over torch.distributed)."""
    steps = 9
    case = _dynamic_case()
    full = _full(case, steps)
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    if how == "file":
        e.comm_init_file(tmp_path / "id", 1, 0, timeout_s=20)
    else:
        e.comm_init(e.comm_unique_id(), 1, 0)
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    ex = run.external_times(1)
    for it in range(1, steps + 1):
        ix = run.internal_times(ex, it)
        e.exchange_rng_window(ex, ix, 1, it)
        e.update_particles(ex, ix, 1, it)
    got = e.get_particles()
    assert np.array_equal(got["par"], full["par"])
    for k in INTS:
        assert np.array_equal(got[k], full[k]), k
    mt, mti = e.get_rng()
    assert mti == full["rng"][1] and np.array_equal(mt, full["rng"][0])
    xyz = e.allgather_positions([case["cfg"].numpar])          # the other collective on the same communicator
    assert np.array_equal(xyz, got["par"][:, :3])
    e.comm_destroy()
    e.close()


def test_exchange_without_a_communicator_and_misuse():
    """no communicator: the engine's own count is the window (a single engine may call it every step); arguments
    that differ from the exchange's are refused by lt_step."""
    case = _dynamic_case()
    full = _full(case, 5)
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    ex = run.external_times(1)
    for it in range(1, 6):
        ix = run.internal_times(ex, it)
        e.exchange_rng_window(ex, ix, 1, it)
        if it == 3:
            with pytest.raises(ltrans_amd.LtError):
                e.update_particles(ex, ix + 1.0, 1, it)
        e.update_particles(ex, ix, 1, it)
    assert np.array_equal(e.get_particles(fields=("par",))["par"], full["par"])
    e.close()


def test_fortran_rank_mode_with_settlement_mortality_and_late_releases(tmp_path):
    """the reference-side host (Fortran, no other library) runs the configuration the static window cannot serve:
    LT_NRANKS=1 with a communicator, habitat polygons through lt_create_polyspecs, lt_exchange_rng_window before
    every lt_step.  Same particles as the Python-driven engine, bit for bit."""
    if not os.path.exists(DRIVER):
        pytest.skip("Fortran harness not built (flang absent at build time)")
    case = _dynamic_case()
    n_ext, n_int = 2, 8
    hab = case["habitat"]
    casefile.write_ltrans_data(tmp_path / "LTRANS.data", case["cfg"], days=1.0, habitat=hab)
    casefile.write_case(tmp_path / "case.bin", case["g"], case["geo"], case["bounds"], case["parts"],
                        [case["record"](r) for r in range(n_ext + 1)], case["cfg"].numpar, habitat=hab)
    env = dict(os.environ, LT_NRANKS="1", LT_RANK="0", LT_COMM_FILE=str(tmp_path / "comm_id"), LT_DEVICE="0")
    r = subprocess.run([DRIVER, str(tmp_path / "LTRANS.data"), str(tmp_path / "case.bin"), str(tmp_path / "out.bin"),
                        str(n_ext), str(n_int), "0"], capture_output=True, text=True, cwd=tmp_path, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    f = casefile.read_result(tmp_path / "out.bin.rank0")
    # the same lists as the harness builds (createPolySpecs on the device), the same loops
    ref_case = dict(case, habitat=None, habitat_raw=hab)
    p = cases.run_case(ref_case, ltrans_amd.Engine, {}, n_external=n_ext, max_internal=n_int)
    assert np.array_equal(f["par"], p["par"])
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand", "dead", "oob", "settle"):
        assert np.array_equal(f[k], p[k]), k
    assert np.array_equal(f["xyz_all"], p["par"][:, :3])
    assert p["settle"].sum() > 0 and p["dead"].sum() >= 0
