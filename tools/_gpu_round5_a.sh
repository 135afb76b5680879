mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/gpu_tests2.log 2>&1; echo rc=$? >> gpurun_out/gpu_tests2.log
tail -15 gpurun_out/gpu_tests2.log
python tools/particles_timing.py > gpurun_out/particles_timing.txt 2>&1
python tools/prep_timing.py > gpurun_out/prep_timing.txt 2>&1
bash tools/ab_env_shapes.sh "DRP_ECACHE_MAX_N=64 DRP_ECACHE_MAX_N=256" "256x80 256x100 256x150 256x200 1024x80 1024x100 1024x128 1024x150 1024x200 1024x256 4096x50 8192x20" > gpurun_out/ab_ecache_n.txt 2>&1
DRP_ECACHE_MAX_N=64 python tools/gd_timing.py 50 64 80 100 > gpurun_out/gd_ec64.txt 2>&1
DRP_ECACHE_MAX_N=256 python tools/gd_timing.py 50 64 80 100 > gpurun_out/gd_ec256.txt 2>&1
python bench.py > gpurun_out/bench_default2.json 2> gpurun_out/bench_default2.err
tail -c 600 gpurun_out/bench_default2.json
