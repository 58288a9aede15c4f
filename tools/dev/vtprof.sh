#!/bin/bash
# usage (on the GPU box, from the repo root): tools/dev/vtprof.sh tag1 tag2 ... -- per-kernel times of the turbulence
# step (bench.py --turb under rocprofv3 --kernel-trace --stats) for each build/lib_<tag>.so
repo=$PWD
for t in "$@"; do
  out=$repo/gpurun_out/vtprof_$t
  mkdir -p $out
  export LT_HIP_LIBRARY=$repo/build/lib_$t.so
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 $repo/bench.py --turb --steps 8 --warmup 3 --no-cpu-baseline --no-hydro-timing --sustained-steps 0 $BENCH_EXTRA > $out/bench.json 2> $out/bench.err) || { echo "$t: failed"; tail -3 $out/bench.err; continue; }
  python3 - "$t" "$out" <<'PY'
import csv, json, sys
t, out = sys.argv[1], sys.argv[2]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("== %s: %.4f ms/step under the profiler, sigerr %s" % (t, d["ms_per_step"], d["config"]["sigerr_last_step"]))
for r in list(csv.DictReader(open(out + "/r_kernel_stats.csv")))[:9]:
    print("   %-52s calls %4s avg %9.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
