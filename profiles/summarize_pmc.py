#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv (two separate passes)
-> per-kernel HBM-side bytes per launch, written to profiles/traffic.json.

Corrections, from /opt/skills/guides/MI355X_MICROARCH.md (section HBM): both counters are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane coalesced
read, WRITE_SIZE is exact for 16-B-per-lane stores.  Calibrated on this code's own access
pattern: km_update reads three [B*N,64] fp32 buffers with float4 loads = 230 400 KiB per
launch at 1024 x 300 and FETCH_SIZE reports 115 461 KiB (x0.501).

usage: summarize_pmc.py <engine | gd-demo> <fetch_counter_collection.csv> <write_counter_collection.csv>
(the per-kernel reductions profiles/r04_<preset>_pmc_{fetch,write}_per_kernel.csv are accepted as well)
"""
import collections
import csv
import json
import os
import re
import sys

CLASS = {'k_graph': 'graph', 'k_graph_strips': 'graph', 'k_graph_sort': 'graph_sort', 'k_node_encode': 'node_encode', 'k_edge_encode': 'edge_encode',
         'k_project': 'project', 'k_aggregate': 'aggregate', 'k_aggregate_lds': 'aggregate', 'k_update': 'update',
         'k_predict': 'predict', 'km_node_encode': 'node_encode', 'km_node_encode_split': 'node_encode', 'km_edge_encode': 'edge_encode',
         'km_update<false>': 'update', 'km_update<true>': 'predict', 'k_reward': 'reward',
         'km_prop<false>': 'prop', 'km_prop<true>': 'prop_last', 'km_prop3': 'prop3',
         'km_rollout': 'rollout', 'k_graph_cells': 'graph', 'k_graph_sort2': 'graph_sort',
         'kmb_step_bwd': 'step_bwd', 'kmb_rows_bwd': 'step_bwd', 'k_graph_rev': 'graph', 'kb_reward': 'bwd_reward', 'kb_reverse_lists': 'bwd_lists', 'kb_sdelta': 'bwd_push'}


def counts_work(name):
    """True for the WORK instantiations (the calibration iteration's counting kernels): km_prop<LAST, TAPE, PAIR, WORK>,
    km_prop3<TAPE, PAIR, ECACHE, WORK, ONE>, km_rollout<PAIR, ECACHE, WORK, ONE>."""
    m = re.match(r'(km_prop3|km_prop|km_rollout)<([^>]*)>', name.strip())
    if not m:
        return False
    a = [x.strip() for x in m.group(2).split(',')]
    i = {'km_prop': 3, 'km_prop3': 3, 'km_rollout': 2}[m.group(1)]
    return len(a) > i and a[i] == 'true'


def norm(name):
    # the counting instantiations (one calibration iteration of bench.py) are not the kernels the timed region runs
    if counts_work(name):
        return '(counting instantiation)'
    # km_prop gained a second template argument (tape); both spellings mean the same kernel here
    if name.startswith('km_prop<false'):
        return 'km_prop<false>'
    if name.startswith('km_prop<true'):
        return 'km_prop<true>'
    if name.startswith('km_prop3'):      # the three propagation steps of a rollout step in one launch
        return 'km_prop3'
    if name.startswith('k_graph_strips'):     # <T>, and _q<T> (quarter-wave ranges, round 3)
        return 'k_graph_strips'
    if name.startswith('km_rollout'):         # <false> / <true> (paired tiles): the whole rollout of a small pile in one launch
        return 'km_rollout'
    return name


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = norm(r['Kernel_Name'].split('(')[0].replace('void ', '').strip())
        if 'Mean_Counter_Value' in r:         # a per-kernel reduction (profiles/reduce_pmc.py) instead of the raw collection
            d[name].extend([float(r['Mean_Counter_Value'])] * int(r['Dispatches']))
        else:
            d[name].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in d.items()}, {k: len(v) for k, v in d.items()}


def main():
    engine, fpath, wpath = sys.argv[1:4]
    if engine == 'gd-demo':
        CLASS['km_prop3'] = 'prop3_tape'      # the tape-writing instantiation km_prop3<true> of the GD planner's forward
    fetch, nf = per_kernel(fpath, 'FETCH_SIZE')
    write, _ = per_kernel(wpath, 'WRITE_SIZE')
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'traffic.json')
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    rows = {}
    for k in sorted(fetch):
        base = k.split('<')[0] if k not in CLASS else k
        cls = CLASS.get(k, CLASS.get(base))
        if cls is None:
            continue
        fb = 2.0 * fetch[k] * 1024.0
        wb = write.get(k, 0.0) * 1024.0
        rows[cls] = {'kernel': k, 'launches_sampled': nf[k], 'fetch_bytes_raw_counter': fetch[k] * 1024.0,
                     'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes_per_launch': fb + wb}
        print('%-22s %-20s fetch %8.1f MB (raw %8.1f)  write %8.1f MB' %
              (cls, k, fb / 1e6, fetch[k] * 1024 / 1e6, wb / 1e6))
    out[engine] = rows
    json.dump(out, open(out_path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
