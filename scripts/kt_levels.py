#!/usr/bin/env python3
"""Per-launch durations of the last step of a bench run from a rocprofv3 kernel trace (CSV): scripts/kt_levels.py DIR"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'prep' in r['Kernel_Name']] or \
    [i for i, r in enumerate(rows) if 'reset_err' in r['Kernel_Name']]  # (a folded sweep has no per-branch pass)
last_prep = max(starts)
tot = {}
for r in rows[last_prep:]:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    tot[name] = tot.get(name, 0.0) + us
    if us > 100:
        print('%-40s %9.1f us' % (name[:40], us))
print({k: round(v, 1) for k, v in tot.items() if v > 20})
