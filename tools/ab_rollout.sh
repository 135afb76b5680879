#!/bin/bash
# km_rollout (the whole rollout in one launch) against the step-by-step pipeline, per particle count
for n in "$@"; do
  for v in "DRP_DUMMY=0" "DRP_NO_ROLLOUT_FUSED=1"; do
    env $v python bench.py --particles $n --samples 1024 --horizon 10 --steps 30 --warmup 5 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N=$n $v  %.3f ms/iter  %.4g  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']), {k: v for k, v in d['kernel_ms_per_iteration'].items() if v})
"
  done
done
