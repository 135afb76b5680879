#!/bin/bash
# A/B of library builds on one box: tools/ab.sh lib1.so lib2.so ...  (each run twice, interleaved)
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DRP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-alt --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], 'ms/iter', d['value'], d.get('roofline', {}).get('frac'), d.get('kernels_ms', ''))
"
  done
done
