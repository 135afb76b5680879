#!/bin/bash
# A/B of library builds on one box at a config, many iterations: tools/ab_long.sh "<bench args>" lib1.so lib2.so ...  (each three times, interleaved)
args="$1"; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    DRP_LIB=$PWD/$lib python bench.py $args --no-cpu-baseline --no-alt --no-sweep --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s %.4f ms/iter  %.4g  median %.4f' % ('$lib', d['ms_per_step'], d['value'], d.get('median_ms_per_step', 0)))
"
  done
done
