#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE against kernels of known traffic (tools/dev/traffic_calib.hip).
#   usage (through gpurun):  bash profiles/tools/calibrate.sh <tag>   -> gpurun_out/<tag>/calib.txt
set -e -o pipefail
tag=${1:-calib}
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
exe=$repo/build/traffic_calib
[ -x "$exe" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$exe" "$repo/tools/dev/traffic_calib.hip" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
"$exe" > "$out/expect.txt"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/f" -o r -- "$exe" > /dev/null 2>"$out/f.err"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/w" -o r -- "$exe" > /dev/null 2>"$out/w.err"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
exp = {}
for ln in open(out + "/expect.txt").read().splitlines()[1:]:
    p = ln.split()
    exp[p[0]] = (int(p[1]), int(p[2]), " ".join(p[3:]))
got = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in "fw":
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            got[k][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["%-26s %14s %14s %8s | %14s %14s %8s  %s" % ("kernel", "read expected", "FETCH_SIZE KiB*1024", "ratio", "write expected", "WRITE_SIZE KiB*1024", "ratio", "")]
for k, (er, ew, note) in exp.items():
    g = got.get(k, {})
    fr, wr = g.get("FETCH_SIZE", float("nan")) * 1024, g.get("WRITE_SIZE", float("nan")) * 1024
    lines.append("%-26s %14d %14.0f %8.3f | %14d %14.0f %8.3f  %s" % (k, er, fr, fr / er if er else float("nan"), ew, wr, wr / ew if ew else float("nan"), note))
open(out + "/calib.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
