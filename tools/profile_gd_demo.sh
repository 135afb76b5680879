# kernel stats of the GD planner's iteration at the demo shape (bench.py --config gd-demo)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gd_demo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gd_demo -- python3 bench.py --config gd-demo --no-cpu-baseline --steps 30 --warmup 3 > gpurun_out/prof_gd_demo.log 2>&1
tail -1 gpurun_out/prof_gd_demo.log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gd_demo/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print('%-46s calls %6s avg %10.1f us  %5s %%' % (r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
