#!/bin/bash
# usage (GPU box, repo root): tools/dev/vtab.sh <rounds> "<ENV=.. ENV=.. tag>" ... -- the turbulence line (bench.py --turb) of each
# variant build/lib_<tag>.so under its environment, in turn, several rounds (compare within a round)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    tag=${v##* }; envs=${v% *}; [ "$envs" == "$v" ] && envs=""
    env $envs LT_HIP_LIBRARY=$PWD/build/lib_$tag.so timeout -k 10 200 python3 bench.py --turb --no-cpu-baseline --no-hydro-timing --steps ${STEPS:-20} --warmup ${WARMUP:-6} $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-58s' % '$v', 'ms/step %.4f' % d['ms_per_step'], 'vt_group_ms %.4f' % d['roofline']['kernel_ms_avg'], 'sigerr', d['config']['sigerr_last_step'], 'err', d['config']['errors_last_step'])"
  done
done
