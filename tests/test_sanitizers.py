"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU code (GPU sanitizers are not
available on the pool):

* the oracle (`make -C oracle asan` -> oracle/libltrans_oracle_san.so): a part of the CPU suite in a child process
  with the sanitizer runtimes preloaded and LT_ORACLE_LIBRARY pointing at that build;
* the host-only index arithmetic of the product library -- createBounds (lt_bounds_host.h: the walk along the
  land / sea edge of the mask, boundary_module.f90:82-1215) and the host sides of lt_locate_particles /
  lt_create_polyspecs (lt_init_host.h) -- as tests/csrc/host_san.cpp built by g++ with the same options: the 400
  random masks of tests/test_create_bounds.py go through it, and its arrays are those of the product library.

    make -C oracle asan && python -m pytest tests/test_sanitizers.py
"""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _clean(stderr):
    return "AddressSanitizer" not in stderr and "runtime error:" not in stderr and "LeakSanitizer" not in stderr


def test_oracle_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan or not shutil.which("gcc"):
        pytest.skip("gcc sanitizer runtimes not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, LT_ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "libltrans_oracle_san.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    # the oracle against the reference's vectors, the engine-level cases (every family: advection, behaviours,
    # turbulence, reflection, settlement, free slip), createBounds, the init-time geometry
    files = ["test_golden.py", "test_edge_cases.py", "test_behaviour.py", "test_freeslip.py", "test_create_bounds.py",
             "test_init_geometry.py", "test_host.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] +
                       [os.path.join(HERE, f) for f in files], env=env, capture_output=True, text=True, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert _clean(r.stderr) and _clean(r.stdout), (r.stdout + r.stderr)[-3000:]
    assert " passed" in r.stdout


def test_host_index_arithmetic_under_asan_and_ubsan(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("g++ not present")
    exe = str(tmp_path / "host_san")
    subprocess.check_call(["g++", "-std=c++17"] + SAN_FLAGS + ["-I", os.path.join(ROOT, "ltransv.2b_amd", "csrc"), "-o", exe,
                           os.path.join(HERE, "csrc", "host_san.cpp")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    # the host sides of lt_locate_particles / lt_create_polyspecs on a small rectilinear case
    r = subprocess.run([exe, "init"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and _clean(r.stderr), r.stderr[-3000:]
    assert "error 3 at 500" in r.stdout                               # the particle without a v element, reported as setEle_all does
    assert r.stdout.strip().endswith("polyspecs 1 5 0 8 4 3 0 4 holes 0 1 1 1 2")
    # createBounds on the random masks of test_create_bounds.py
    sys.path.insert(0, HERE)
    import test_create_bounds as tcb
    import ltrans_amd
    import oracle_lib
    rng = np.random.default_rng(20260101)
    vi, uj = 30, 22
    masks = []
    for _ in range(400):
        sides = "".join(s for s in "nsew" if rng.random() < 0.3)
        m = tcb._random_mask(rng, vi, uj, p_land=rng.uniform(0.15, 0.5), open_sides=sides)
        if m is not None:
            masks.append(np.asfortranarray(m.astype(np.int32)))
    assert len(masks) > 300
    with open(tmp_path / "masks.bin", "wb") as f:
        f.write(struct.pack("<3i", len(masks), vi, uj))
        for m in masks:
            f.write(m.tobytes(order="F"))
    r = subprocess.run([exe, "bounds", str(tmp_path / "masks.bin"), str(tmp_path / "out.bin")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and _clean(r.stderr), r.stderr[-3000:]
    # ... and the arrays are the product library's (same header, compiled by hipcc into libltrans_hip.so)
    ui, vj = vi - 1, uj - 1
    ii, jj = np.meshgrid(np.arange(ui), np.arange(uj), indexing="ij")
    g = dict(vi=vi, uj=uj, x_u=ii + 0.5, y_u=jj.astype(float))
    ii, jj = np.meshgrid(np.arange(vi), np.arange(vj), indexing="ij")
    g.update(x_v=ii.astype(float), y_v=jj + 0.5, mask_rho=masks[0])
    hip = ltrans_amd.load_hip_library()
    buf = open(tmp_path / "out.bin", "rb").read()
    off, n_ok = 0, 0
    for m in masks:
        rc, nb, mb, mi, ni = struct.unpack_from("<5i", buf, off)
        off += 20
        try:
            want = oracle_lib.create_bounds(hip, "lt_", g, m)
        except RuntimeError:
            want = None
        assert (rc != 0) == (want is None)
        if rc != 0:
            continue
        n_ok += 1
        assert (nb, mb, mi, ni) == (want["nbounds"], want["maxbound"], want["maxisland"], want["numislands"])

        def take(dtype, count):
            nonlocal off
            a = np.frombuffer(buf, dtype=dtype, count=count, offset=off)
            off += a.nbytes
            return a
        assert np.array_equal(take("<f8", mb), want["bx"]) and np.array_equal(take("<f8", mb), want["by"])
        assert np.array_equal(take("<f8", mi), want["hx"]) and np.array_equal(take("<f8", mi), want["hy"])
        assert np.array_equal(take("<i4", mi), want["hid"]) and np.array_equal(take("<i4", nb), want["land"])
        assert np.array_equal(take("<f8", 2 * nb).reshape(2, nb, order="F"), want["bnd_x"])
        assert np.array_equal(take("<f8", 2 * nb).reshape(2, nb, order="F"), want["bnd_y"])
    assert off == len(buf) and n_ok > 200
