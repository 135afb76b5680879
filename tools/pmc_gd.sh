# counters of the GD planner's two big kernels at the demo shape (kmb_step_bwd, km_prop3<TAPE>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gd1 gpurun_out/pmc_gd2 gpurun_out/pmc_gd3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_gd1 -- python3 bench.py --config gd-demo --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_gd2 -- python3 bench.py --config gd-demo --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc_gd3 -- python3 bench.py --config gd-demo --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for dname in ('pmc_gd1', 'pmc_gd2', 'pmc_gd3'):
    fs = glob.glob('gpurun_out/%s/*/*_counter_collection.csv' % dname)
    if not fs: print(dname, 'no output'); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in d:
        if 'kmb_step_bwd' in k or 'km_prop3' in k: print(dname, k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d[k].items()})
PY
find gpurun_out/pmc_gd1 gpurun_out/pmc_gd2 gpurun_out/pmc_gd3 -name "*.csv" -size +5M -delete
