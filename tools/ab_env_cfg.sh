#!/bin/bash
# A/B of environment switches at given particle counts on one box: tools/ab_env_cfg.sh "VAR=1" N1 N2 ...
V=$1; shift
for n in "$@"; do
  for v in "DRP_DUMMY=0" "$V"; do
    env $v python bench.py --particles $n --samples 1024 --horizon 10 --steps 20 --warmup 3 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N=$n $v  %.3f ms/iter  %.4g' % (d['ms_per_step'], d['value']), {k: v for k, v in d['kernel_ms_per_iteration'].items() if v})
"
  done
done
