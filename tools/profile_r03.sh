# Round-3 profiles of the bench command (run on the GPU box through gpurun; outputs under gpurun_out/r03/, the
# reduced summaries under gpurun_out/r03/summ are then copied to profiles/r03_*).  Counters are collected in their
# own passes, one rocprofv3 process each, the program itself (python3) after `--`.  The profiled command is the
# headline workload alone (--no-sweep): a kernel's average over the sweep's other shapes would not be the number the
# bench line's roofline is computed from.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
rm -rf $O && mkdir -p $O/summ
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16"
SQ2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"
run() {  # tag, bench arguments...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -- python3 bench.py "$@" > $O/${tag}_stats.log 2>&1
  cp $(ls $O/${tag}_stats/*/*_kernel_stats.csv | head -1) $O/summ/r03_${tag}_kernel_stats.csv
}
pmc() {  # tag, pass name, counters (one string), bench arguments...
  tag=$1; pass=$2; ctr=$3; shift 3
  rocprofv3 --pmc $ctr --output-format csv -d $O/${tag}_${pass} -- python3 bench.py "$@" > $O/${tag}_${pass}.log 2>&1
  python3 profiles/reduce_pmc.py $(ls $O/${tag}_${pass}/*/*_counter_collection.csv | head -1) $O/summ/r03_${tag}_pmc_${pass}_per_kernel.csv
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
cp profiles/traffic.json $O/summ/traffic.json
# the sweep's workloads: kernel stats and HBM-side bytes of their dominant kernel (traffic.json entries under the preset's name)
for c in p20 c4-50 c4-150 c4-600 c5-share; do
  S="--config $c --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-sweep"
  run $c $S
  pmc $c fetch FETCH_SIZE $S
  pmc $c write WRITE_SIZE $S
  python3 profiles/summarize_pmc.py $c $(ls $O/${c}_fetch/*/*_counter_collection.csv | head -1) $(ls $O/${c}_write/*/*_counter_collection.csv | head -1) > $O/summ/traffic_$c.txt
done
cp profiles/traffic.json $O/summ/traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 tools/train_timing.py > $O/train_stats.log 2>&1
cp $(ls $O/train_stats/*/*_kernel_stats.csv | head -1) $O/summ/r03_train_kernel_stats.csv
# the bench lines of this build on this box
python3 bench.py > $O/summ/r03_bench_default.json 2> $O/bench_default.err
python3 bench.py --config gd-demo > $O/summ/r03_bench_gd_demo.json 2>/dev/null
# keep only the summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
du -sh $O
