#!/bin/bash
# 1/2/4/8-GPU series of the two sharded BASELINE workloads on one node (the driver runs the default series itself;
# this is for a maintainer with an 8-GPU box).  One process per GPU, RCCL over xGMI, rendezvous on 127.0.0.1.
#   c3: weak scaling, 1024 samples per GPU (8192 at 8 GPUs = BASELINE configs[2])
#   c5: strong scaling, 4096 samples in total (512 per GPU at 8 GPUs = BASELINE configs[4])
# usage: [GPU_COUNTS="1 2 4 8"] tools/scale.sh [steps] [warmup]     -> one JSON line per run in scale_<config>.jsonl
set -e
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-3}; PORT=${MASTER_PORT:-29621}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for CFG in c3 c5; do
  : > scale_$CFG.jsonl
  for N in ${GPU_COUNTS:-1 2 4 8}; do
    if [ "$N" = 1 ]; then
      python bench.py --gpus 1 --config $CFG --steps $STEPS --warmup $WARM --no-alt | tail -1 >> scale_$CFG.jsonl
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --config $CFG --steps $STEPS --warmup $WARM --no-alt | tail -1 >> scale_$CFG.jsonl
    fi
  done
  python - "$CFG" <<'PY'
import json, sys
rows = [json.loads(l) for l in open('scale_%s.jsonl' % sys.argv[1]) if l.strip().startswith('{')]
base = rows[0]['value']
for r in rows:
    print('%s  %d GPU(s)  %-6s  %.3e particle-steps/s  %.2f ms/iteration  x%.2f' % (
        sys.argv[1], r['n_gpus'], r['scaling'], r['value'], r['ms_per_step'], r['value'] / base))
PY
done
