mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/gpu_tests4.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests4.log
tail -8 gpurun_out/gpu_tests4.log
python tools/prep_timing.py > gpurun_out/prep_timing4.txt 2>&1
