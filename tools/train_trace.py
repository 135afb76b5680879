"""Post-process a rocprofv3 --kernel-trace of tools/train_timing.py (see tools/train_trace.sh)."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2] == 'k_adam']
# update iterations: from after the re-pack that follows one k_adam to the next k_adam's re-pack; take iterations 5..29
cuts = adam[4:30]
busy = defaultdict(float); gap = defaultdict(float); calls = defaultdict(int)
n_it = len(cuts) - 1
tot = 0.0
for a, b in zip(cuts[:-1], cuts[1:]):
    tot += rows[b][0] - rows[a][0]
    for i in range(a, b):
        s, e, n = rows[i]
        busy[n] += e - s; calls[n] += 1
        gap[n] += max(0, rows[i + 1][0] - e)
print('%d iterations, %.1f us each (k_adam to k_adam on the device clock), %.1f launches' % (n_it, tot / n_it / 1e3, sum(calls.values()) / n_it))
print('%-44s %6s %9s %9s' % ('kernel', 'calls', 'busy us', 'gap-after us'))
for n in sorted(busy, key=lambda n: -(busy[n] + gap[n])):
    print('%-44s %6.1f %9.1f %9.1f' % (n[:44], calls[n] / n_it, busy[n] / n_it / 1e3, gap[n] / n_it / 1e3))
print('%-44s %6s %9.1f %9.1f' % ('total', '', sum(busy.values()) / n_it / 1e3, sum(gap.values()) / n_it / 1e3))
# the sequence of one iteration
a, b = cuts[-2], cuts[-1]
print('\nlast iteration, in order (start us, busy us, gap-after us):')
for i in range(a, b):
    s, e, n = rows[i]
    print('%8.1f %7.1f %7.1f  %s' % ((s - rows[a][0]) / 1e3, (e - s) / 1e3, (rows[i + 1][0] - e) / 1e3, n[:60]))
