cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gd -- python3 tools/gd_timing.py > gpurun_out/prof_gd.log 2>&1
tail -4 gpurun_out/prof_gd.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_gd/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('%-40s calls %6s avg %10.1f us  %5s %%' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
