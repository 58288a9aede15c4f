import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import cases, ltrans_amd, oracle_lib
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_reflect"
ora = cases.run_case(cases.build_case(name), ltrans_amd.Engine, dict(lib=oracle_lib.load_oracle(), prefix="lto_"), n_external=2)
for every in (-1, 1, 7):
    out = cases.run_case(cases.build_case(name, resort_every=every), ltrans_amd.Engine, {}, n_external=2)
    d = np.abs(out["par"][:, :3] - ora["par"][:, :3]).max(axis=1)
    bad = np.flatnonzero(d > 0)
    print("resort_every", every, "particles differing from the oracle:", bad.size, "max abs", d.max(), bad[:10], d[bad[:10]])
    print("   err", np.bincount(out["err_code"]), "stats", out["stats"])
