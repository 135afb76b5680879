python bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-sweep > /dev/null 2>&1
for sh in ${SHAPES:-256x100 1024x100 256x80 1024x80 512x128}; do
  s=${sh%x*}; n=${sh#*x}
  for rep in 1 2; do
    for mx in 64 ${MAXN:-128}; do
      DRP_ROLLOUT_MAX_N=$mx python bench.py --particles $n --samples $s --horizon 10 --steps 100 --warmup 20 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%5s x %-4s max_n %-4s %.4f ms/iter  %.4g' % ('$s', '$n', '$mx', d['ms_per_step'], d['value']))
"
    done
  done
done
