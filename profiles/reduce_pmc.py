#!/usr/bin/env python3
"""A rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter, 0.3 - 2 MB) -> one row per kernel and
counter: dispatches, mean, min, max.   usage: reduce_pmc.py <counter_collection.csv> <out.csv>"""
import collections, csv, sys

d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].split('(')[0].replace('void ', '').strip(), r['Counter_Name'])
    d.setdefault(k, []).append(float(r['Counter_Value']))
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kernel_Name', 'Counter_Name', 'Dispatches', 'Mean_Counter_Value', 'Min', 'Max'])
    for (k, c), v in d.items():
        w.writerow([k, c, len(v), '%g' % (sum(v) / len(v)), '%g' % min(v), '%g' % max(v)])
