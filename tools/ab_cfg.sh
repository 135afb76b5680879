#!/bin/bash
# A/B of library builds on one box over several workloads: tools/ab_cfg.sh "cfgA cfgB ..." lib1.so lib2.so ...
# (a library path of "-" means the in-tree libdrp.so); each run twice, interleaved
CFGS="$1"; shift
for cfg in $CFGS; do
  for rep in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = "-" ]; then L=""; else L="$PWD/$lib"; fi
      DRP_LIB=$L python bench.py --config $cfg --no-cpu-baseline --no-alt --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-8s %-22s %.3f ms (median %.3f)  %.4e  prop %.3f' % ('$cfg', '$lib', d['ms_per_step'], d['ms_per_step_median'], d['value'], d['kernel_ms_per_iteration'].get('prop', 0)))
"
    done
  done
done
