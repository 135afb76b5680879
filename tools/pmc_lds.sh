cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds gpurun_out/pmc_wait
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_wait -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for dname in ('pmc_lds', 'pmc_wait'):
    fs = glob.glob('gpurun_out/%s/*/*_counter_collection.csv' % dname)
    if not fs: print(dname, 'no output'); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in d:
        if 'km_prop3' in k or 'graph_strips' in k: print(dname, k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d[k].items()})
PY
