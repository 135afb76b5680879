# Where one training iteration's time goes at batch 4 x <=300 particles: the kernel trace of tools/train_timing.py's first
# shape, cut into iterations at k_adam; per kernel the busy time and the idle gap that FOLLOWS it on the stream.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_train
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_train -- python3 tools/train_timing.py 0 30 > gpurun_out/trace_train.log 2>&1
cat gpurun_out/trace_train.log | tail -3
python3 tools/train_trace.py gpurun_out/trace_train > gpurun_out/train_trace.txt
cat gpurun_out/train_trace.txt
