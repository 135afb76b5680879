# kernel trace of the GD planner's iteration at small piles: kernel time per iteration against the wall time
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in ${GD_NS:-20 50}; do
rm -rf gpurun_out/prof_gd_$n
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gd_$n -- python3 tools/gd_timing.py $n > gpurun_out/prof_gd_$n.log 2>&1
tail -1 gpurun_out/prof_gd_$n.log
python3 - $n <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_gd_%s/*/*_kernel_stats.csv' % sys.argv[1])[0]
tot = 0.0
for r in list(csv.DictReader(open(f))):
    calls = int(r['Calls'])
    if calls >= 23:
        print('%-40s calls %6s avg %10.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
        tot += float(r['AverageNs']) / 1e3 * (calls // 23)
print('sum of kernel time per iteration: %.1f us' % tot)
PY
done
