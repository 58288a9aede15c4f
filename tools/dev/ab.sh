#!/bin/bash
# usage (GPU box): tools/dev/ab.sh <rounds> "<ENV=.. tag>" "<ENV=.. tag>" ...  -- the bench line of each variant, in turn, several
# rounds (clock and temperature drift between runs: compare within a round)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    tag=${v##* }; envs=${v% *}; [ "$envs" == "$v" ] && envs=""
    env $envs LT_HIP_LIBRARY=$PWD/build/lib_$tag.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-hydro-timing --no-turbulence-line --steps ${STEPS:-30} --warmup ${WARMUP:-10} $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-32s' % '$v', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'err', d['config']['errors_last_step'])"
  done
done
