cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_graph gpurun_out/prof_graph_sq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python3 bench.py --particles 1200 --samples 512 --horizon 20 --steps 5 --warmup 2 --no-alt --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof_graph_sq -- python3 bench.py --particles 1200 --samples 512 --horizon 20 --steps 3 --warmup 1 --no-alt --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_graph/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print('%-46s calls %6s avg %10.1f us  %5s %%' % (r['Name'][:46], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
f = glob.glob('gpurun_out/prof_graph_sq/*/*_counter_collection.csv')[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
    d[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in d:
    if 'graph' in k: print(k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d[k].items()})
PY
find gpurun_out/prof_graph gpurun_out/prof_graph_sq -name "*trace*" -delete
