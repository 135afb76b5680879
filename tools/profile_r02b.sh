cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b
rm -rf $O && mkdir -p $O
for c in c2 c4-50 c4-600 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 bench.py --config $c --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $O/${c}_stats.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gd_stats -- python3 bench.py --config gd-demo --steps 30 --warmup 3 --no-cpu-baseline > $O/gd_stats.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 bench.py > $O/bench_default.json 2> /dev/null
python3 bench.py --config gd-demo > $O/bench_gd_demo.json 2>/dev/null
python3 bench.py --config c4-50 > $O/bench_c4-50.json 2>/dev/null
python3 bench.py --particles 1200 --samples 512 --horizon 20 --no-alt > $O/bench_c5share.json 2>/dev/null
python3 bench.py --config c4-600 --no-alt --no-cpu-baseline > $O/bench_c4-600.json 2>/dev/null
python3 bench.py --config c4-150 --no-alt --no-cpu-baseline > $O/bench_c4-150.json 2>/dev/null
