"""bench.py's own multi-rank launch: `python bench.py --gpus N` with no launcher around it must start
N ranks itself (before anything touches the GPU), relay rank 0's line and fail when a rank fails; a
WORLD_SIZE that disagrees with --gpus is an error.  CPU part: the launcher alone (gloo rendezvous of the
children).  GPU part: the 2-rank rehearsal of the benchmark itself on one card (LT_BENCH_BACKEND=gloo:
the ranks share the GPU, the all-gather goes through host memory), on a small grid with the structure
of BASELINE configs[3] (turbulence, strong scaling, the shards' windows of the serial random stream)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _json_line(stdout):
    """the WHOLE standard output is one JSON line (library banners of any rank go to stderr: bench.claim_stdout)"""
    assert stdout.endswith("\n") and stdout.count("\n") == 1, stdout[:2000]
    return json.loads(stdout)


def test_launcher_starts_n_ranks_and_relays_rank0():
    r = _run(["--gpus", "2", "--selftest-launch", "ok"])
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["sum_ok"] is True


def test_under_torchrun_the_whole_stdout_is_one_json_line():
    """the way the driver starts N > 1: python -m torch.distributed.run ... bench.py --gpus N.  The ranks share the
    launcher's stdout; gloo's "[Gloo] Rank r is connected ..." banner of every rank must not land in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2",
                        "--selftest-launch", "ok"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["sum_ok"] is True


def test_launcher_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--selftest-launch", "fail"])
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--selftest-launch", "ok"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_workload_resolution():
    """what `bench.py --gpus N` measures: the N = 1 configuration per GPU (weak scaling, one series with N = 1) unless a
    workload is named; BASELINE configs[3] strong-scaled as the embedded line of an auto run"""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    one = bench.resolve_workload(a, 1)
    assert (one["name"], one["total"], one["weak"], one["turb"], one["grid"]) == ("config2", 1_000_000, False, False, "512x256x30")
    eight = bench.resolve_workload(a, 8)
    assert (eight["name"], eight["total"], eight["weak"], eight["grid"]) == ("config2", 8_000_000, True, "512x256x30")
    emb = bench.resolve_workload(a, 8, name="config4")
    assert (emb["name"], emb["total"], emb["weak"], emb["turb"], emb["grid"]) == ("config4", 10_000_000, False, True, "1024x512x50")
    named = bench.resolve_workload(bench.parse(["--workload", "config4", "--gpus", "8"]), 8)
    assert (named["name"], named["total"], named["weak"]) == ("config4", 10_000_000, False)
    small = bench.parse(["--gpus", "2", "--grid", "128x64x20", "--numpar", "20000", "--configs3-grid", "64x32x10",
                         "--configs3-numpar", "5000"])
    assert bench.resolve_workload(small, 2)["total"] == 40000 and bench.resolve_workload(small, 2)["grid"] == "128x64x20"
    e2 = bench.resolve_workload(small, 2, name="config4")
    assert (e2["total"], e2["grid"], e2["weak"]) == (5000, "64x32x10", False)
    turb = bench.resolve_workload(bench.parse(["--turb"]), 1)
    assert turb["name"] == "config3" and turb["turb"]
    # bytes and time of a roofline figure cover the same kernels
    r = bench.roofline_block("f32", True, 31, 1_000_000, 4.0e-3, 5.3e-3, None, None, None)
    assert r["algorithmic_bytes_per_particle_step"] == 48 * 31 + 992 and r["kernel"].count("+") == 4
    assert abs(r["frac"] - 2480e6 / 4.0e-3 / 1e9 / 8000.0) < 1e-12
    assert r["whole_step"]["algorithmic_bytes_per_particle_step"] == 976 + 48 * 31 + 992


@pytest.mark.gpu
def test_two_rank_rehearsal_of_the_benchmark():
    r = _run(["--gpus", "2", "--workload", "config4", "--grid", "128x64x20", "--numpar", "40000", "--steps", "3",
              "--warmup", "1"], {"LT_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert d["config"]["particles_total"] == 40000 and d["config"]["particles_this_rank"] == 20000
    assert d["config"]["baseline_config"] == "config4" and d["config"]["rng_window"].startswith("static")
    assert d["config"]["collective_backend"] == "gloo" and "gloo" in d["config"]["workload"]
    assert d["config"]["errors_last_step"] == 0 and d["value"] > 0
    assert d["config"]["single_gpu_same_workload"]["value"] > 0
    assert d["cpu_baseline"] is None


@pytest.mark.gpu
def test_two_rank_default_line_is_the_headline_workload_weak_scaled():
    """`bench.py --gpus N` as the driver runs it: the configuration of the N = 1 line per GPU (weak scaling, one series
    with N = 1), and BASELINE configs[3] strong-scaled with its one-GPU point under config.configs3_strong."""
    r = _run(["--gpus", "2", "--grid", "128x64x20", "--numpar", "20000", "--configs3-grid", "128x64x20",
              "--configs3-numpar", "40000", "--steps", "3", "--warmup", "1"], {"LT_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["baseline_config"] == "config2"
    assert d["config"]["particles_total"] == 40000 and d["config"]["particles_this_rank"] == 20000
    assert d["roofline"]["kernel"] == "k_rk" and d["config"]["rng_window"] is None and d["value"] > 0
    c3 = d["config"]["configs3_strong"]
    assert c3["scaling"] == "strong" and c3["particles_total"] == 40000 and c3["particles_this_rank"] == 20000
    assert c3["rng_window"].startswith("static") and c3["errors_last_step"] == 0 and c3["value"] > 0
    assert c3["single_gpu_same_workload"]["value"] > 0
    assert d["config"]["total_s"] > 0 and d["cpu_baseline"] is None


@pytest.mark.gpu
def test_single_gpu_line_carries_config3_and_hash():
    r = _run(["--steps", "3", "--warmup", "1", "--numpar", "20000", "--grid", "128x64x20", "--no-cpu-baseline",
              "--no-hydro-timing"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["baseline_config"] == "config2"
    t = d["config"]["turbulence"]
    assert t["value"] > 0 and t["roofline"]["kernel"] == "k_vt_table+k_vt_sigs+k_vt_norm+k_vt_walk_fast+k_vt_walk"
    # bytes and time of a figure cover the same kernels: the VTurb group against A_turb - A_adv, the step against A_turb
    assert t["roofline"]["algorithmic_bytes_per_particle_step"] == 48 * 21 + 992
    assert t["roofline"]["whole_step"]["algorithmic_bytes_per_particle_step"] == 976 + 48 * 21 + 992
    assert t["rng_window"].startswith("static") and t["create_to_first_step_s"] > 0
    assert d["config"]["resorts_in_timed_region"] == 0 and d["config"]["total_s"] > d["config"]["setup_s"] > 0
    # no re-sort fell into the 3 timed steps: the line says what the window left out (3/32 of a re-sort)
    assert d["config"]["resort_ms_each"] > 0
    assert abs(d["config"]["resort_s_excluded"] - 3 / 32 * d["config"]["resort_ms_each"] * 1e-3) < 1e-12
    assert len(d["config"]["kernel_source_hash"]) == 16
    assert d["roofline"]["traffic"] is None          # not the profiled shape


@pytest.mark.gpu
def test_two_rank_rehearsal_under_torchrun():
    """the way the driver starts it: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", BENCH, "--gpus", "2", "--workload", "config4",
                        "--grid", "128x64x20", "--numpar", "40000", "--steps", "3", "--warmup", "1", "--no-single-ref"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["particles_this_rank"] == 20000
    assert d["config"]["errors_last_step"] == 0 and d["value"] > 0
