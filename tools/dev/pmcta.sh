#!/bin/bash
# usage (GPU box): build/pmcta.sh <tag> <lib tag> -- TA / TCP counters of k_rk (few per pass)
tag=$1; lib=$2
repo=$PWD
export LT_HIP_LIBRARY=$repo/build/lib_$lib.so
out=$repo/gpurun_out/pmcta_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" "TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2 TCP_TOTAL_ACCESSES"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 > $out/p$i.log 2>&1 || echo "pass $i failed: $set"
  echo "pass $i done"
done
cd $repo
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rk<" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, c) in sorted(acc.items()):
    print(k, "rows %d mean/launch %.5g" % (c, s / max(c, 1)))
PY
