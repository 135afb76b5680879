mkdir -p gpurun_out
python tools/_dbg_bwd.py > gpurun_out/dbg_bwd.txt 2>&1
python -m pytest tests/test_gpu_fuzz_oracle.py tests/test_gpu_gd.py tests/test_gpu_planner.py tests/test_gpu_fullsize.py tests/test_gpu_goal.py tests/test_gpu_parity.py tests/test_gpu_bench.py -q -s -p no:cacheprovider > gpurun_out/gpu_tests3.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests3.log
tail -12 gpurun_out/gpu_tests3.log
python tools/prep_timing.py > gpurun_out/prep_timing3.txt 2>&1
bash tools/ab_env_shapes.sh "- DRP_ECACHE_MAX_N=64" "256x100 256x240 1024x100 1024x240 1024x150" > gpurun_out/ab_ecache_n3.txt 2>&1
python tools/gd_timing.py 20 40 50 64 100 > gpurun_out/gd_timing3.txt 2>&1
