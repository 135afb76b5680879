cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01_fused9 --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-alt --no-cpu-baseline > gpurun_out/prof_fused9.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcn_fetch --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-alt --no-cpu-baseline > gpurun_out/pmcn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmcn_write --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-alt --no-cpu-baseline > gpurun_out/pmcn_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 -d gpurun_out/pmcn_sq --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-alt --no-cpu-baseline > gpurun_out/pmcn_sq.log 2>&1
tail -1 gpurun_out/prof_fused9.log | cut -c1-300
find gpurun_out/prof_r01_fused9 -name "*kernel_stats*" | head
python3 bench.py 2>&1 | tail -1 > gpurun_out/bench_r01_final.json; cut -c1-200 gpurun_out/bench_r01_final.json
