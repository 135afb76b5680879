# environment settings of the library over shapes on one box, interleaved, after a warm-up:
#   tools/ab_env_shapes.sh "VAR=a VAR=b ..." "S1xN1 S2xN2 ..."     (a setting "-" = the defaults)
python bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-sweep > /dev/null 2>&1
for sh in $2; do
  s=${sh%x*}; n=${sh#*x}
  for rep in 1 2; do
    for kv in $1; do
      if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
      env $pre python bench.py --particles $n --samples $s --horizon 10 --steps 150 --warmup 30 --no-alt --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%5s x %-4s %-28s %.4f ms/iter  %.4g' % ('$s', '$n', '$kv', d['ms_per_step'], d['value']))
"
    done
  done
done
