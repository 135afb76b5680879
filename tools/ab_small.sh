#!/bin/bash
# A/B of library builds at small piles on one box: tools/ab_small.sh "N1 N2 ..." lib1.so lib2.so ...  (each twice, interleaved)
sizes="$1"; shift
for n in $sizes; do
  for rep in 1 2; do
    for lib in "$@"; do
      DRP_LIB=$PWD/$lib python bench.py --particles $n --samples ${SAMPLES:-1024} --horizon 10 --steps 40 --warmup 5 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N=%-4s %-28s %.4f ms/iter  %.4g' % ('$n', '$lib', d['ms_per_step'], d['value']))
"
    done
  done
done
