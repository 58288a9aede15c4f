#!/bin/bash
# usage (on the GPU box): build/benchlibs.sh tag1 tag2 ... ; bench each build/lib_<tag>.so
for t in "$@"; do
  LT_HIP_LIBRARY=$PWD/build/lib_$t.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-hydro-timing --steps 30 --warmup 5 $BENCH_EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'sigerr', d['config']['sigerr_last_step'], 'err', d['config']['errors_last_step'])"
done
