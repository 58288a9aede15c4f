#!/bin/bash
# usage (GPU box): build/pmcx.sh <tag> <lib tag> -- extra SQ counters of k_rk, 2 passes
tag=$1; lib=$2
repo=$PWD
export LT_HIP_LIBRARY=$repo/build/lib_$lib.so
out=$repo/gpurun_out/pmcx_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $out/a -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 > $out/a.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $out/b -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 > $out/b.log 2>&1
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $out/c -o r -- python3 $repo/bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps 4 --warmup 2 > $out/c.log 2>&1
cd $repo
python3 - <<PY
import csv, glob, collections
for sub in "abc":
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rk<" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    # rows are per dispatch (x per-XCD?) : normalise per launch by SQ_WAVES count of rows
    for k, (s, c) in sorted(acc.items()):
        print(sub, k, "sum %.4g rows %d mean %.5g" % (s, c, s / max(c, 1)))
PY
