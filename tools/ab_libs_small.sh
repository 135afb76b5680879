#!/bin/bash
# library builds over small shapes on one box, interleaved, after a warm-up run (a cold GPU clocks low for the first
# tens of milliseconds: a 30-ms bench of a small shape would measure that):  tools/ab_libs_small.sh "lib1 lib2" "S1xN1 S2xN2 ..."
libs="$1"; shapes="$2"
python bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-sweep > /dev/null 2>&1
for sh in $shapes; do
  s=${sh%x*}; n=${sh#*x}
  for rep in 1 2 3; do
    for lib in $libs; do
      DRP_LIB=$PWD/$lib python bench.py --particles $n --samples $s --horizon 10 --steps 60 --warmup 20 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%5s x %-4s %-34s %.4f ms/iter  %.4g' % ('$s', '$n', '$lib', d['ms_per_step'], d['value']))
"
    done
  done
done
