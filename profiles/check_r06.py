#!/usr/bin/env python3
"""Round 5: every `frac` of the driver line's sweep block against the counters on file.  The bench line's numerator is what
the kernels counted themselves (drp_probe_work); here the same quantity comes from the SQ pass of the same preset:
(SQ_INSTS_VALU_MFMA_MOPS_F16 + _BF16) / 64 = 16-bit MFMA instructions per launch of the dominant kernel.
usage: python3 profiles/check_r06.py   (reads profiles/r06_bench_default.json and profiles/r06_<preset>_pmc_sq_per_kernel.csv)"""
import csv
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def counts_work(name):
    """True for the WORK instantiations (the calibration iteration's counting kernels): km_prop<LAST, TAPE, PAIR, WORK>,
    km_prop3<TAPE, PAIR, ECACHE, WORK, ONE>, km_rollout<PAIR, ECACHE, WORK, ONE>."""
    m = re.match(r'(km_prop3|km_prop|km_rollout)<([^>]*)>', name.strip())
    if not m:
        return False
    a = [x.strip() for x in m.group(2).split(',')]
    i = {'km_prop': 3, 'km_prop3': 3, 'km_rollout': 2}[m.group(1)]
    return len(a) > i and a[i] == 'true'


def mfmas_from_sq(tag, prefixes):
    best = None
    for r in csv.DictReader(open(os.path.join(HERE, 'r06_%s_pmc_sq_per_kernel.csv' % tag))):
        name = r['Kernel_Name']
        if not name.startswith(prefixes) or counts_work(name) or not r['Counter_Name'].startswith('SQ_INSTS_VALU_MFMA_MOPS'):
            continue
        best = best or {}
        best.setdefault(name, 0.0)
        best[name] += float(r['Mean_Counter_Value']) / 64.0
    return best


line = [json.loads(l) for l in open(os.path.join(HERE, 'r06_bench_default.json')) if l.startswith('{')][0]
rows = [('fused', line['roofline'], ('km_prop3',))]
for e in line['sweep']:
    if e['name'] == 'gd-demo':
        continue
    rows.append((e['name'], e, ('km_rollout', 'km_prop3')))
print('%-10s %-44s %14s %14s %8s' % ('preset', 'dominant kernel (SQ pass)', 'SQ MFMAs', 'counted MFMAs', 'ratio'))
for tag, e, pre in rows:
    sq = mfmas_from_sq(tag, pre)
    name = max(sq, key=sq.get)
    counted = e['executed_per_launch']['mfmas']
    print('%-10s %-44s %14.0f %14.0f %8.4f' % (tag, name[:44], sq[name], counted, counted / sq[name]))
