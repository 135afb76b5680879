#!/bin/bash
# Do the dispatch thresholds (csrc/capi_ctx.h: which kernel serves which shape) still pick the faster side ON THIS BOX?
# At each boundary the default and the forced alternative run interleaved, twice, on the shape just inside and just outside it.
#   bash tools/validate_dispatch.sh            (about 3 minutes of GPU time; prints ms per MPPI iteration, lower is better)
# The thresholds were tuned on pool boxes that differ by their sustained clock, not by relative kernel costs;
# a line where the alternative wins by more than a few per cent on a new part or driver is the one to re-tune.
run() {  # samples particles "ENV=..."|-
  if [ "$3" = "-" ]; then pre=""; else pre="$3"; fi
  env $pre python bench.py --particles $2 --samples $1 --horizon 10 --steps 120 --warmup 30 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('    %5s x %-4s %-26s %.4f ms/iter  %s' % ('$1', '$2', '$3', d['ms_per_step'], d['roofline']['kernel']))
"
}
ab() {  # title samples particles alternative
  echo "== $1"
  for rep in 1 2; do run $2 $3 -; run $2 $3 "$4"; done
}
python bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-sweep > /dev/null 2>&1      # clocks up
ab "whole rollout in one launch up to 64 particles: 1024 x 64, default (km_rollout) vs step pipeline"      1024 64  DRP_NO_ROLLOUT_FUSED=1
ab "... and not above (workgroups over 256 rows): 1024 x 80, default (km_prop3) vs km_rollout"             1024 80  DRP_ROLLOUT_MAX_N=128
ab "km_rollout for small workgroups up to 256 particles: 256 x 150, default (km_rollout) vs step pipeline" 256 150  DRP_NO_ROLLOUT_FUSED=1
ab "edge-chain cache up to 128 particles: 1024 x 128, default (cached) vs recomputing"                     1024 128 DRP_ECACHE_MAX_MB=0
ab "... and not from 129: 1024 x 150, default (recomputing) vs cached"                                     1024 150 DRP_ECACHE_MAX_N=256
ab "... but from 225 to 256: 1024 x 240, default (cached) vs recomputing"                                  1024 240 DRP_ECACHE_MAX_MB=0
ab "x strips from 129 particles: 1024 x 150, default (k_graph_strips_q) vs plain k_graph"                  1024 150 DRP_NO_GRAPH_STRIPS=1
ab "two-dimensional cells from 400: 1024 x 400, default (k_graph_cells) vs x strips"                       1024 400 DRP_NO_GRAPH_CELLS=1
ab "... and not below: 1024 x 350, default (x strips) vs cells"                                            1024 350 DRP_GRAPH_CELLS_MIN_N=300
ab "three propagation steps in one launch: 1024 x 300, default (km_prop3) vs one km_prop launch per step" 1024 300 DRP_NO_PROP3=1
ab "paired tiles up to 192 rows per workgroup: 1024 x 32, default (pair) vs tiles of 32"                   1024 32  DRP_PROP_PAIR_ROWS=0
