#!/bin/bash
# one library, one environment switch, a list of (samples particles) shapes: tools/ab_env_small.sh VAR "v1 v2 ..." "S1xN1 S2xN2 ..."
var="$1"; vals="$2"; shapes="$3"
for sh in $shapes; do
  s=${sh%x*}; n=${sh#*x}
  for rep in 1 2; do
    for v in $vals; do
      env $var=$v python bench.py --particles $n --samples $s --horizon 10 --steps 40 --warmup 5 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%5s x %-4s %s=%-5s %.4f ms/iter  %.4g  in-degree %s' % ('$s', '$n', '$var', '$v', d['ms_per_step'], d['value'], d.get('config', {}).get('mean_in_degree', d.get('mean_in_degree'))))
"
    done
  done
done
