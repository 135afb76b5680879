#!/bin/bash
# km_rollout against the step-by-step pipeline for small BATCHES: tools/ab_rollout_batch.sh N samples...
n=$1; shift
for b in "$@"; do
  for v in "DRP_DUMMY=0" "DRP_NO_ROLLOUT_FUSED=1"; do
    env $v python bench.py --particles $n --samples $b --horizon 10 --steps 30 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N=$n B=$b $v  %.3f ms/iter  %.4g' % (d['ms_per_step'], d['value']))
"
  done
done
