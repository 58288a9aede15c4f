#!/bin/bash
# usage (GPU box, repo root): tools/dev/tcp_counters.sh <tag> ... -- vector-memory counters of the turbulence step's
# kernels (bench.py --turb, 5 steps) for each build/lib_<tag>.so: L1 requests / L2 requests / L2 hits and misses
repo=$PWD
for t in "$@"; do
  export LT_HIP_LIBRARY=$repo/build/lib_$t.so
  for grp in "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum"; do
    out=$repo/gpurun_out/tcp_$t/$(echo $grp | cut -c1-7)
    mkdir -p $out
    (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o r -- python3 $repo/bench.py --turb --steps 5 --warmup 2 --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 > $out/log 2>&1) || { echo "$t $grp: failed"; tail -3 $out/log; }
  done
  python3 - "$t" "$repo/gpurun_out/tcp_$t" <<'PY'
import csv, glob, sys, collections
t, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_vt_" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", t)
for k in sorted(acc):
    print("  %-28s" % k[:28], "  ".join("%s %.3g" % (c.replace("_sum", "").replace("TCP_", "").replace("TCC_", "L2_"), sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
done
