"""The C-ABI boundary: the HIP library loads (no GPU needed), exports every
symbol include/ltrans_hip.h declares, and the ctypes mirror matches the header's
struct layout (checked with a C probe compiled from the header)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import ltrans_amd
from ltrans_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ltrans_hip.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lt_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_hip()
    lib = ltrans_amd.load_hip_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ltrans_hip.h but not exported"
    lib.lt_abi_version.restype = C.c_int
    assert lib.lt_abi_version() == abi.LT_ABI_VERSION


def test_python_entry_points_are_declared():
    names = set(declared_functions())
    for n in abi.ENTRY_POINTS + abi.HIP_ONLY_ENTRY_POINTS:
        assert "lt_" + n in names


def test_oracle_exports_same_engine_api(oracle):
    for n in abi.ENTRY_POINTS:
        assert hasattr(oracle, "lto_" + n)


STRUCTS = {"lt_config": abi.LtConfig, "lt_grid": abi.LtGrid, "lt_bounds": abi.LtBounds,
           "lt_habitat": abi.LtHabitat, "lt_particles": abi.LtParticles, "lt_slab": abi.LtSlab,
           "lt_step_stats": abi.LtStepStats, "lt_rom_record": abi.LtRomRecord,
           "lt_output": abi.LtOutput}


def test_struct_layout_matches_header(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in STRUCTS.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in STRUCTS.items():
        assert int(out[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_create_rejects_bad_abi_and_unsupported_options():
    lib = ltrans_amd.load_hip_library()
    h = C.c_void_p()
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9)
    cfg.struct_size = 12
    assert lib.lt_create(C.byref(cfg), C.byref(h)) == -2          # LT_ERR_ABI
    for kw in (dict(Behavior=8), dict(Behavior=-1)):
        cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9, **kw)
        assert lib.lt_create(C.byref(cfg), C.byref(h)) == -4      # LT_ERR_UNSUPPORTED
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=10)
    assert lib.lt_create(C.byref(cfg), C.byref(h)) == -1          # ws must be us+1
    # a shard's draw offsets are 32-bit: 17.4M particles x 124 draws would wrap
    cfg = ltrans_amd.LtConfig(numpar=17_400_000, us=8, ws=9, idt=120, HTurbOn=1, VTurbOn=1)
    assert lib.lt_create(C.byref(cfg), C.byref(h)) == -1


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the engine must fail loudly (LT_ERR_DEVICE), not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = ltrans_amd.LtConfig(numpar=4, us=8, ws=9)
    with pytest.raises(ltrans_amd.LtError) as ei:
        ltrans_amd.Engine(cfg)
    assert ei.value.code == -5


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ltrans_amd.load_hip_library(str(tmp_path / "nope.so"))
