#!/usr/bin/env python3
"""Generate tests/golden/traj_golden.npz: whole-run trajectories of the CPU ORACLE on the tiny
synthetic cases of tests/cases.py (the reference itself cannot be built here, DESIGN.md
section 2, so these vectors pin the ORACLE + HIP path against regressions between rounds, not
the oracle against the reference).  Data only: final particle state after a fixed number of
steps, per case."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import oracle_lib  # noqa: E402
import ltrans_amd  # noqa: E402

CASES = {  # name: (n_external, max_internal)
    "tiny_adv": (3, 6), "tiny_sink": (2, 6), "tiny_vt2": (2, 4), "tiny_vt3": (2, 4), "tiny_turb": (2, 4),
    "tiny_reflect": (2, 8), "tiny_open": (2, 8), "tiny_settle": (2, None), "tiny_beh2": (2, 5),
    "tiny_beh3": (2, 5), "tiny_beh5": (3, 5), "tiny_beh5_old": (3, 5), "tiny_beh7": (3, 5), "tiny_salt": (2, 4),
}
KEEP = ("par", "P_r_element", "P_behave", "dead", "oob", "settle", "endpoly", "hitBottom", "hitLand",
        "err_code", "timer", "bottom", "P_Salt")


def run(name, kw):
    ne, mi = CASES[name]
    return cases.run_case(cases.build_case(name), ltrans_amd.Engine, kw, n_external=ne, max_internal=mi)


if __name__ == "__main__":
    kw = dict(lib=oracle_lib.load_oracle(), prefix="lto_")
    out = {}
    for name in CASES:
        r = run(name, kw)
        for k in KEEP:
            out[f"{name}/{k}"] = r[k]
        print(name, "ok")
    np.savez_compressed(os.path.join(HERE, "traj_golden.npz"), **out)
