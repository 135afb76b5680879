cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcx_fetch
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcx_fetch -- python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline > gpurun_out/pmcx_fetch.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmcx_fetch/*/*_counter_collection.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'].split('(')[0].replace('void ','').strip()].append(float(r['Counter_Value']))
for k in sorted(d):
    if 'km_' in k or 'k_graph' in k: print(k, len(d[k]), 'fetch %.1f MB (x2 corrected)' % (2*sum(d[k])/len(d[k])*1024/1e6))
PY
