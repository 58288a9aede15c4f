#!/bin/bash
# How many host synchronisations does a turbulence step cost?  Two runs of bench.py --turb that differ only in the
# number of timed steps, each under rocprofv3 --hip-runtime-trace --stats (no counters: API tracing and --pmc are never
# combined); the difference of the hipStreamSynchronize / hipEventSynchronize / hipMemcpy call counts, divided by the
# extra steps, is what one step costs.  (Neither window contains a slab rotation: 5 + 10 and 5 + 20 steps of 30.)
#   usage (through gpurun):  bash profiles/tools/api_sync.sh <tag> [bench.py args; default --turb]   -> gpurun_out/<tag>/hip_api_sync.txt
set -e -o pipefail
tag=$1; shift
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(--turb)
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for k in 10 20; do
  timeout -k 10 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d "$out/api$k" -o r -- python3 "$repo/bench.py" "${ARGS[@]}" --steps $k --warmup 5 --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 > "$out/api$k.log" 2>&1
done
python3 - "$out" "${ARGS[*]}" <<'PY'
import csv, glob, sys
out = sys.argv[1]
def counts(k):
    c = {}
    for f in glob.glob(out + "/api%d/**/*hip_api_stats.csv" % k, recursive=True) + glob.glob(out + "/api%d/*hip_api_stats.csv" % k):
        for r in csv.DictReader(open(f)):
            c[r["Name"]] = int(r["Calls"])
    return c
a, b = counts(10), counts(20)
names = ["hipStreamSynchronize", "hipEventSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipMemcpyAsync", "hipLaunchKernel", "hipEventRecord", "hipStreamWaitEvent"]
with open(out + "/hip_api_sync.txt", "w") as fh:
    fh.write("bench.py %s, HIP API calls under rocprofv3 --hip-runtime-trace --stats\n" % sys.argv[2])
    fh.write("%-24s %10s %10s %14s\n" % ("call", "10 steps", "20 steps", "per extra step"))
    for n in names:
        fh.write("%-24s %10d %10d %14.2f\n" % (n, a.get(n, 0), b.get(n, 0), (b.get(n, 0) - a.get(n, 0)) / 10.0))
print(open(out + "/hip_api_sync.txt").read())
PY
