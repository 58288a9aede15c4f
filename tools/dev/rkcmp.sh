#!/bin/bash
# usage (GPU box, repo root): tools/dev/rkcmp.sh tag1 tag2 ...
# For each build/lib_<tag>.so: the advection bench line (step time, live k_rk time), then one --pmc pass
# (instruction counts per wave of the step's kernels).  BENCH_EXTRA adds bench.py arguments (e.g. --turb).
repo=$PWD
for t in "$@"; do
  export LT_HIP_LIBRARY=$repo/build/lib_$t.so
  out=$repo/gpurun_out/rkcmp_$t
  mkdir -p $out
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 30 --warmup 5 $BENCH_EXTRA 2>$out/bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'sigerr', d['config']['sigerr_last_step'], 'err', d['config']['errors_last_step'])" || { echo "$t: bench failed"; tail -3 $out/bench.err; continue; }
  (cd /tmp && TMPDIR=/tmp timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/pmc -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 $BENCH_EXTRA > $out/pmc.log 2>&1) || { echo "$t: pmc failed"; tail -3 $out/pmc.log; continue; }
  python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ltd::k_" not in k or "scatter" in k:
            continue
        acc[k.split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    w = max(c["SQ_WAVES"], 1.0)
    cyc = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("   %-40s VALU/wave %8.0f SALU/wave %7.0f SMEM/wave %5.0f  VALU active %.1f %% of wave cycles" % (
        k, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_SMEM"] / w, 100.0 * c["SQ_ACTIVE_INST_VALU"] / cyc))
PY
done
