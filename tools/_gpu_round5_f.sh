mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/gpu_tests6.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests6.log
tail -6 gpurun_out/gpu_tests6.log
python __graft_entry__.py --smoke > gpurun_out/smoke6.txt 2>&1; tail -2 gpurun_out/smoke6.txt
python bench.py > gpurun_out/bench_default6.json 2> gpurun_out/bench_default6.err; tail -c 300 gpurun_out/bench_default6.json
