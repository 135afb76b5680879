mkdir -p gpurun_out
python -m pytest tests/test_gpu_planner.py tests/test_gpu_goal.py tests/test_gpu_fuzz_oracle.py -q -s -p no:cacheprovider -k "fps or goal or preprocessing" > gpurun_out/gpu_tests5.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests5.log
tail -5 gpurun_out/gpu_tests5.log
python tools/prep_timing.py > gpurun_out/prep_timing5.txt 2>&1
