#!/bin/bash
# usage (GPU box, repo root): tools/dev/ab_any.sh <rounds> "<bench args>" "<ENV=.. tag>" ... -- bench.py with the given
# arguments for each variant build/lib_<tag>.so under its environment, in turn, several rounds
rounds=$1; shift
bargs=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    tag=${v##* }; envs=${v% *}; [ "$envs" == "$v" ] && envs=""
    env $envs LT_HIP_LIBRARY=$PWD/build/lib_$tag.so timeout -k 10 300 python3 bench.py $bargs --no-cpu-baseline --no-hydro-timing --no-turbulence-line --sustained-steps 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-40s' % '$v', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'active', d['config']['active_last_step'], 'sigerr', d['config']['sigerr_last_step'], 'err', d['config']['errors_last_step'])"
  done
done
