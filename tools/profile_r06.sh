# Round-6 profiles of the bench command (run on the GPU box through gpurun; outputs under gpurun_out/r06/, the reduced
# summaries under gpurun_out/r06/summ are then copied to profiles/r06_*).  Counters are collected in their own passes, one
# rocprofv3 process each, the program itself (python3) after `--`.  Since round 4: the SQ pass (MFMA / VALU counters) for
# EVERY preset of the sweep block, so that each `frac` of the driver line can be recomputed from profiles/; and the
# FETCH_SIZE calibration on a gather (tools/gather_calib.hip).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
rm -rf $O && mkdir -p $O/summ
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16"
SQ2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"
run() {  # tag, bench arguments...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -- python3 bench.py "$@" > $O/${tag}_stats.log 2>&1
  cp $(ls $O/${tag}_stats/*/*_kernel_stats.csv | head -1) $O/summ/r06_${tag}_kernel_stats.csv
}
pmc() {  # tag, pass name, counters (one string), bench arguments...
  tag=$1; pass=$2; ctr=$3; shift 3
  rocprofv3 --pmc $ctr --output-format csv -d $O/${tag}_${pass} -- python3 bench.py "$@" > $O/${tag}_${pass}.log 2>&1
  python3 profiles/reduce_pmc.py $(ls $O/${tag}_${pass}/*/*_counter_collection.csv | head -1) $O/summ/r06_${tag}_pmc_${pass}_per_kernel.csv
}
B="--steps 10 --warmup 2 --no-alt --no-cpu-baseline --no-sweep"
run fused $B
pmc fused fetch FETCH_SIZE $B
pmc fused write WRITE_SIZE $B
pmc fused sq "$SQ" $B
python3 profiles/summarize_pmc.py fused $(ls $O/fused_fetch/*/*_counter_collection.csv | head -1) $(ls $O/fused_write/*/*_counter_collection.csv | head -1) > $O/summ/traffic_fused.txt
M="--steps 10 --warmup 2 --engine mfma --no-alt --no-cpu-baseline --no-sweep"
run mfma $M
pmc mfma fetch FETCH_SIZE $M
pmc mfma write WRITE_SIZE $M
python3 profiles/summarize_pmc.py mfma $(ls $O/mfma_fetch/*/*_counter_collection.csv | head -1) $(ls $O/mfma_write/*/*_counter_collection.csv | head -1) > $O/summ/traffic_mfma.txt
G="--config gd-demo --steps 30 --warmup 3 --no-cpu-baseline"
run gd_demo $G
pmc gd_demo fetch FETCH_SIZE $G
pmc gd_demo write WRITE_SIZE $G
pmc gd_demo sq "$SQ" $G
pmc gd_demo sq2 "$SQ2" $G
python3 profiles/summarize_pmc.py gd-demo $(ls $O/gd_demo_fetch/*/*_counter_collection.csv | head -1) $(ls $O/gd_demo_write/*/*_counter_collection.csv | head -1) > $O/summ/traffic_gd.txt
# the sweep's workloads: kernel stats, HBM-side bytes and the MFMA / VALU counters of their dominant kernel
for c in p20 c4-50 c4-150 c4-600 c5-share; do
  S="--config $c --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-sweep"
  run $c $S
  pmc $c fetch FETCH_SIZE $S
  pmc $c write WRITE_SIZE $S
  pmc $c sq "$SQ" $S
  python3 profiles/summarize_pmc.py $c $(ls $O/${c}_fetch/*/*_counter_collection.csv | head -1) $(ls $O/${c}_write/*/*_counter_collection.csv | head -1) > $O/summ/traffic_$c.txt
done
cp profiles/traffic.json $O/summ/traffic.json
# FETCH_SIZE on a gather of random 256-B rows (16 B per lane) and on an in-order stream of the same 1 GiB table
hipcc --offload-arch=gfx950 -O3 -o $O/gather_calib tools/gather_calib.hip > $O/gather_calib_build.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $O/gather_calib 1024 > $O/calib.log 2>&1
python3 profiles/reduce_pmc.py $(ls $O/calib_fetch/*/*_counter_collection.csv | head -1) $O/summ/r06_gather_calibration_pmc_fetch.csv
python3 - <<'PY' > gpurun_out/r06/summ/r06_gather_calibration.txt
import csv
known = 1024 << 20
print('FETCH_SIZE calibration (tools/gather_calib.hip, 1 GiB table, every 256-B row read once with 16-B-per-lane loads):')
for r in csv.DictReader(open('gpurun_out/r06/summ/r06_gather_calibration_pmc_fetch.csv')):
    kib = float(r['Mean_Counter_Value'])
    print('  %-16s FETCH_SIZE %.0f KiB per launch = %.4f of the %d KiB read -> correction factor %.3f' %
          (r['Kernel_Name'], kib, kib * 1024 / known, known // 1024, known / (kib * 1024)))
for l in open('gpurun_out/r06/calib.log'):
    if l.startswith('k_'):
        print('  ' + l.strip())
PY
# training: the reference's batch alone (shape 0 of tools/train_timing.py: 2 + 30 update and 2 + 10 forward-only iterations -- the
# launches of an iteration can be counted off the stats), its iterations in order (tools/train_trace.py), then all three shapes unprofiled
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 tools/train_timing.py 0 30 > $O/train_stats.log 2>&1
cp $(ls $O/train_stats/*/*_kernel_stats.csv | head -1) $O/summ/r06_train_kernel_stats.csv
python3 tools/train_trace.py $O/train_stats > $O/summ/r06_train_trace.txt 2>&1
python3 tools/train_timing.py 2>/dev/null | grep "ms per training" > $O/summ/r06_train_timing.txt
python3 tools/particles_timing.py > $O/summ/r06_particles_timing.txt 2>&1
# from the -DROLLOUT_STAMPS build of the same tree (python __graft_entry__.py --lib tools/bin/libdrp_ts.so -DROLLOUT_STAMPS), when it is there
if [ -f tools/bin/libdrp_ts.so ]; then
  DRP_LIB=tools/bin/libdrp_ts.so python3 tools/train_stamps.py 0 > $O/summ/r06_train_stamps.txt 2>&1
  DRP_LIB=tools/bin/libdrp_ts.so python3 tools/rollout_stamps.py 20 > $O/summ/r06_rollout_stamps_20.txt 2>&1
  DRP_LIB=tools/bin/libdrp_ts.so python3 tools/rollout_stamps.py 50 > $O/summ/r06_rollout_stamps_50.txt 2>&1
fi
python3 tools/prep_timing.py > $O/summ/r06_prep_timing.txt 2>&1
# the parity census as the test prints it (tests/test_gpu_census.py)
python3 -m pytest tests/test_gpu_census.py -q -s 2>&1 | grep -v "^\.$" | grep "census\|before it\|after it\|deviation after\|final reward\|passed\|failed" | cut -c1-700 > $O/summ/r06_census.txt
python3 tools/gd_timing.py 5 10 20 30 40 50 100 > $O/summ/r06_gd_timing.txt 2>&1
python3 tools/planner_timing.py > $O/summ/r06_planner_timing.txt 2>&1
# the bench lines of this build on this box
python3 bench.py > $O/summ/r06_bench_default.json 2> $O/bench_default.err
python3 bench.py --config gd-demo > $O/summ/r06_bench_gd_demo.json 2>/dev/null
# keep only the summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
rm -f $O/gather_calib
du -sh $O
