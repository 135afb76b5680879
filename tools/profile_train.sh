cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_timing.py > gpurun_out/prof_train.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_train/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print('%-46s calls %6s avg %10.1f us  %5s %%' % (r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
