# per-config throughput and kernel breakdown of the fused engine (BASELINE configs 2, 4, 5)
for cfg in "50 1024 10" "150 1024 10" "300 1024 10" "600 1024 10" "1200 512 20"; do
  set -- $cfg
  python bench.py --particles $1 --samples $2 --horizon $3 --steps 5 --warmup 2 --no-alt --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('N=%4d ns=%4d H=%2d: %.3e particle-steps/s  %.2f ms/iter  kernels %s  in-degree %.2f' % ($1, $2, $3, d['value'], d['ms_per_step'], {k: v for k, v in d['kernel_ms_per_iteration'].items() if v > 0}, d['config']['mean_in_degree']))"
done
