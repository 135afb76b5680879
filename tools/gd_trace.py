"""Post-process a rocprofv3 --kernel-trace of tools/gd_timing.py N: the kernels of one GD-planner iteration in stream order, busy
time and the idle gap that follows each (is the loop bound by launches or by the kernels?).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gd_trace -- python3 tools/gd_timing.py 20 ; python3 tools/gd_trace.py gpurun_out/gd_trace"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith('kb_sdelta')]          # the iteration's last launch (push gradient + Adam)
cuts = marks[60:200]
busy = defaultdict(float); gap = defaultdict(float); calls = defaultdict(int)
n_it = len(cuts) - 1
tot = 0.0
for a, b in zip(cuts[:-1], cuts[1:]):
    tot += rows[b][1] - rows[a][1]
    for i in range(a + 1, b + 1):
        s, e, n = rows[i]
        busy[n] += e - s; calls[n] += 1
        gap[n] += max(0, rows[i][0] - rows[i - 1][1])
print('%d iterations, %.1f us each (end of kb_sdelta to end of kb_sdelta on the device clock), %.1f launches' % (n_it, tot / n_it / 1e3, sum(calls.values()) / n_it))
print('%-44s %6s %9s %9s' % ('kernel', 'calls', 'busy us', 'gap-before us'))
for n in sorted(busy, key=lambda n: -(busy[n] + gap[n])):
    print('%-44s %6.1f %9.1f %9.1f' % (n[:44], calls[n] / n_it, busy[n] / n_it / 1e3, gap[n] / n_it / 1e3))
print('%-44s %6s %9.1f %9.1f' % ('total', '', sum(busy.values()) / n_it / 1e3, sum(gap.values()) / n_it / 1e3))
