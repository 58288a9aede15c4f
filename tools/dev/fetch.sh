#!/bin/bash
# usage (GPU box): build/fetch.sh libtag... : FETCH_SIZE of k_rk per launch (MB, x2 corrected)
repo=$PWD
for t in "$@"; do
  out=$repo/gpurun_out/fetch_$t; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp LT_HIP_LIBRARY=$repo/build/lib_$t.so timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 $BENCH_EXTRA > $out/log 2>&1)
  python3 - $out $t <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "ltd::" in k: print(sys.argv[2], k[:40], "fetch MB/launch %.1f" % (sum(d.values()) / len(d) * 1024 * 2 / 1e6))
PY
done
