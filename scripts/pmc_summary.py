#!/usr/bin/env python3
"""Per-dispatch SQ counter summary of the big F81 launches: scripts/pmc_summary.py gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b"""
import collections
import csv
import sys

for d in sys.argv[1:]:
    rows = list(csv.DictReader(open(d + '/run_counter_collection.csv')))
    disp = collections.OrderedDict()
    for r in rows:
        dd = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ', '')[:34],
                                                     'grid': int(r['Grid_Size'])})
        dd[r['Counter_Name']] = dd.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    for k, v in disp.items():
        if ('f81_kernel' in v['name'] or 'super_kernel' in v['name']) and v.get('SQ_WAVES', v.get('SQ_WAVE_CYCLES', 0)) > 0:
            big = {a: round(b / 1e6, 1) for a, b in v.items() if a not in ('name', 'grid')}
            if max(big.values()) > 100:
                print(k, v['name'], v['grid'], big)
