#!/usr/bin/env python3
"""Static instruction mix of k_rk between the `; MARK n` comments of a -DLT_RK_MARK build
(development aid).  usage: isa_sections.py file.s [kernel-substring]"""
import collections
import re
import sys

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_rk"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3ltd\d+" + kern + r"I.*:", l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
sec = "pre"
cnt = collections.OrderedDict()
for l in lines[start:end]:
    m = re.search(r"; MARK (\d+)", l)
    if m:
        sec = "after%s" % m.group(1)
        continue
    m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)", l)
    if not m:
        continue
    op = m.group(1)
    d = cnt.setdefault(sec, collections.Counter())
    d[op] += 1
for sec, d in cnt.items():
    valu = sum(v for k, v in d.items() if k.startswith("v_"))
    salu = sum(v for k, v in d.items() if k.startswith("s_"))
    mem = sum(v for k, v in d.items() if not k[0:2] in ("v_", "s_"))
    f64 = sum(v for k, v in d.items() if "f64" in k and k.startswith("v_"))
    rw = d["v_readlane_b32"] + d["v_writelane_b32"]
    mov = d["v_mov_b32_e32"] + 2 * d["v_mov_b64_e32"] + d["v_accvgpr_write_b32"] + d["v_accvgpr_read_b32"]
    cnd = sum(v for k, v in d.items() if k.startswith("v_cndmask"))
    print(f"{sec:10s} valu={valu:5d} (f64={f64:4d} cndmask={cnd:4d} mov={mov:4d} lane_rw={rw:3d}) salu={salu:4d} mem={mem:3d}")
    if "-v" in sys.argv:
        print("   ", ", ".join(f"{k}:{v}" for k, v in d.most_common(25)))
