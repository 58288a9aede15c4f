"""ltransv.2b_amd/csrc/lt_libm.h compiled for the host (the same source the kernels compile) against
the host libm: exp, log, log10, pow(x, 0.5), cos, sin, sincos, atan must return identical bits on millions
of arguments.  This pins the restatement (operation order, FMA placement, tables) without a GPU; the
device build is checked by tests/test_gpu_libm.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lt_libm_matches_host_libm(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    with open("/proc/cpuinfo") as f:
        if " fma " not in f.read():
            pytest.skip("host CPU without FMA: its libm runs other variants than the ones restated")
    exe = str(tmp_path / "libm_check")
    subprocess.check_call([gcc, "-O2", "-mfma", "-ffp-contract=off", "-std=gnu11", "-I",
                           os.path.join(ROOT, "ltransv.2b_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "libm_check.c"), "-o", exe, "-lm"])
    out = subprocess.check_output([exe, "1500000"]).decode().split("\n")
    seen = {}
    for line in out:
        if line.strip():
            name, bad, n = line.split()
            seen[name] = (int(bad), int(n))
    assert set(seen) == {"exp", "log", "log10", "pow_half", "cos", "sin", "sincos", "atan"}
    wrong = {k: v for k, v in seen.items() if v[0] != 0}
    if wrong:
        # every function wrong by a little = another libm; a single one wrong = a bug here
        if len(wrong) == len(seen):
            pytest.xfail(f"the host libm is not glibc >= 2.28 (FMA variants): {wrong}")
        raise AssertionError(f"lt_libm.h differs from the host libm: {wrong}")
