python bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-sweep > /dev/null 2>&1
for rep in 1 2 3 4 5 6; do
  for lib in ab/libdrp_fill.so ab/libdrp_ovl3.so; do
    DRP_LIB=$PWD/$lib python bench.py --particles 20 --samples 1024 --horizon 10 --steps 400 --warmup 50 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-22s %.4f ms/iter (median %.4f)  %.4g' % ('$lib', d['ms_per_step'], d['ms_per_step_median'], d['value']))
"
  done
done
