#!/bin/bash
# A/B of environment switches on one box: tools/ab_env.sh "VAR=1" ...   (each run twice, interleaved with the default)
run() { env $1 python bench.py --no-cpu-baseline --no-alt --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.3f ms/iter' % d['ms_per_step'], '%.4g' % d['value'], d['kernel_ms_per_iteration'])
"; }
for rep in 1 2; do
  echo "== default"; run "DRP_DUMMY=0"
  for v in "$@"; do echo "== $v"; run "$v"; done
done
