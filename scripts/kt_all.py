#!/usr/bin/env python3
"""All launches of the last repetition in a rocprofv3 kernel trace: scripts/kt_all.py DIR [marker-kernel-substring]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'pij'
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if marker in r['Kernel_Name'])
tot = {}
t_first = int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    tot[name] = tot.get(name, 0.0) + us
    print('%-46s %9.1f us  at %9.1f  grid %s' % (name[:46], us, (int(r['Start_Timestamp']) - t_first) / 1e3, r['Grid_Size_X']))
print({k: round(v, 1) for k, v in tot.items()})
