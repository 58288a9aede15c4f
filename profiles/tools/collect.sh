#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py configuration on the GPU box.
#   usage (through gpurun):  bash profiles/tools/collect.sh <tag> [bench.py args...]
# Writes everything under gpurun_out/<tag>/ (scratch) and a small summary under
# gpurun_out/<tag>/summary/, which is what gets copied into profiles/<tag>/.
#  pass 1: --kernel-trace --stats            -> kernel_stats.csv (per-kernel durations)
#  pass 2: --pmc <8 SQ counters>             -> issue / wait / occupancy picture
#  pass 3: --pmc FETCH_SIZE                  -> bytes read at the L2 memory side
#  pass 4: --pmc WRITE_SIZE                  -> bytes written
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass)
# and never combined with hip/hsa tracing.
set -e -o pipefail
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$repo/gpurun_out/$tag
mkdir -p "$out/summary"
cd /tmp && export TMPDIR=/tmp
run() {  # <subdir> <rocprof args...> -- bench args
  local sub=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$out/$sub" -o r -- python3 "$repo/bench.py" "${BENCH_ARGS[@]}" --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 > "$out/$sub.log" 2>&1
  grep '^{' "$out/$sub.log" > "$out/$sub.bench.json" || true
}
BENCH_ARGS=("$@")
run stats --kernel-trace --stats
BENCH_ARGS=("$@" --steps 5 --warmup 2)
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
python3 "$repo/profiles/tools/summarize.py" "$out"
cat "$out/summary/summary.txt"
