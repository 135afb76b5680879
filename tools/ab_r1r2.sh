cd $GRAFT_REPO_ROOT
for cfg in "300 1024 10" "20 1024 10" "50 1024 10" "150 1024 10" "600 1024 10" "1200 512 20" "100 128 10"; do
  set -- $cfg
  for rep in 1 2; do
    (cd ab/r1 && python bench.py --particles $1 --samples $2 --horizon $3 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r1 %s x %s x %s: %.3f ms  %.4e  graph %.3f prop %.3f' % ('$1','$2','$3', d['ms_per_step'], d['value'], d['kernel_ms_per_iteration']['graph'], d['kernel_ms_per_iteration']['prop']))")
    python bench.py --particles $1 --samples $2 --horizon $3 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r2 %s x %s x %s: %.3f ms (median %.3f)  %.4e  graph %.3f prop %.3f' % ('$1','$2','$3', d['ms_per_step'], d['ms_per_step_median'], d['value'], d['kernel_ms_per_iteration']['graph'], d['kernel_ms_per_iteration']['prop']))"
  done
done
(cd ab/r1 && python tools/gd_timing.py 2>/dev/null | sed 's/^/r1 /' | cut -c1-70)
python tools/gd_timing.py 2>/dev/null | sed 's/^/r2 /' | cut -c1-70
