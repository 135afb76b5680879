# Round-2 profiles of the bench command (run on the GPU box through gpurun; outputs under gpurun_out/r02/, the
# summaries are then copied to profiles/r02_* by hand).  Counters are collected in their own passes, one
# rocprofv3 process each, the program itself (python3) after `--`.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 10 --warmup 2 --no-alt --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused_stats -- $B > $O/fused_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fused_fetch -- $B > $O/fused_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/fused_write -- $B > $O/fused_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/fused_sq -- $B > $O/fused_sq.log 2>&1
M="python3 bench.py --steps 10 --warmup 2 --engine mfma --no-alt --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfma_stats -- $M > $O/mfma_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mfma_fetch -- $M > $O/mfma_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/mfma_write -- $M > $O/mfma_write.log 2>&1
G="python3 bench.py --config gd-demo --steps 30 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gd_stats -- $G > $O/gd_stats.log 2>&1
for c in c4-50 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 bench.py --config $c --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $O/${c}_stats.log 2>&1
done
# keep only the summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --config gd-demo > $O/bench_gd_demo.json 2>/dev/null
du -sh $O
