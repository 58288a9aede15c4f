#!/bin/bash
# usage (GPU box, repo root): tools/dev/kprof.sh "<bench args>" tag1 tag2 ... -- per-kernel times of bench.py with the
# given arguments under rocprofv3 --kernel-trace --stats, for each build/lib_<tag>.so
repo=$PWD
bargs=$1; shift
for t in "$@"; do
  out=$repo/gpurun_out/kprof_$t
  mkdir -p $out
  export LT_HIP_LIBRARY=$repo/build/lib_$t.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 $repo/bench.py $bargs --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 > $out/bench.json 2> $out/bench.err) || { echo "$t: failed"; tail -3 $out/bench.err; continue; }
  python3 - "$t" "$out" <<'PY'
import csv, json, sys
t, out = sys.argv[1], sys.argv[2]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("== %s: %.4f ms/step under the profiler" % (t, d["ms_per_step"]))
for r in list(csv.DictReader(open(out + "/r_kernel_stats.csv")))[:10]:
    if r["Name"].startswith("Cijk"): continue
    print("   %-52s calls %4s avg %9.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
