#!/bin/bash
# How many host synchronisations does a turbulence step cost?  Two runs of bench.py --turb that differ only in the
# number of timed steps, each under rocprofv3 --hip-runtime-trace --stats (no counters: API tracing and --pmc are never
# combined); the difference of the hipStreamSynchronize / hipEventSynchronize / hipMemcpy call counts, divided by the
# extra steps, is what one step costs.  (K1, K2 = the two step counts, default 10 and 20: neither window then contains a
# slab rotation.  A workload without a static window: take K1 = 40, K2 = 70 -- bench.py runs on until a re-sort has
# happened, which equalises runs shorter than 32 steps.)
#   usage (through gpurun):  bash profiles/tools/api_sync.sh <tag> [bench.py args; default --turb]   -> gpurun_out/<tag>/hip_api_sync.txt
set -e -o pipefail
tag=$1; shift
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(--turb)
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
K1=${K1:-10}; K2=${K2:-20}
for k in $K1 $K2; do
  timeout -k 10 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d "$out/api$k" -o r -- python3 "$repo/bench.py" "${ARGS[@]}" --steps $k --warmup 5 --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 > "$out/api$k.log" 2>&1
done
python3 - "$out" "${ARGS[*]}" $K1 $K2 <<'PY'
import csv, glob, sys
out = sys.argv[1]
def counts(k):
    c = {}
    for f in glob.glob(out + "/api%d/**/*hip_api_stats.csv" % k, recursive=True) + glob.glob(out + "/api%d/*hip_api_stats.csv" % k):
        for r in csv.DictReader(open(f)):
            c[r["Name"]] = int(r["Calls"])
    return c
K1, K2 = int(sys.argv[3]), int(sys.argv[4])
a, b = counts(K1), counts(K2)
names = ["hipStreamSynchronize", "hipEventSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipMemcpyAsync", "hipLaunchKernel", "hipEventRecord", "hipStreamWaitEvent"]
with open(out + "/hip_api_sync.txt", "w") as fh:
    fh.write("bench.py %s, HIP API calls under rocprofv3 --hip-runtime-trace --stats\n" % sys.argv[2])
    fh.write("%-24s %10s %10s %14s\n" % ("call", "%d steps" % K1, "%d steps" % K2, "per extra step"))
    for n in names:
        fh.write("%-24s %10d %10d %14.2f\n" % (n, a.get(n, 0), b.get(n, 0), (b.get(n, 0) - a.get(n, 0)) / float(K2 - K1)))
print(open(out + "/hip_api_sync.txt").read())
PY
