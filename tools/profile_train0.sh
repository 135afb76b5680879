cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_train0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train0 -- python3 tools/train_timing.py 0 20 > gpurun_out/prof_train0.log 2>&1
grep "ms per" gpurun_out/prof_train0.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_train0/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
iters = 34.0   # 2 + 20 update, 2 + 10 eval
tot = 0
for r in rows[:40]:
    per = float(r['TotalDurationNs']) / 1e3 / 22.0
    print('%-60s calls %5s avg %7.1f us   total/22 %7.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, per))
print('sum over all kernels / 22 update iterations: %.1f us (includes the 12 eval iterations)' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e3 / 22.0))
PY
