#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --pmc ... --kernel-trace run: scripts/r04_prof_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, os, re, sys

def short(name):
    name = name.replace('void ', '')
    m = re.match(r'([A-Za-z0-9_]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:40]

for d in sys.argv[1:]:
    paths = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)
    if not paths:
        print(d, ': no counter file'); continue
    path = max(paths, key=os.path.getmtime)
    acc = collections.OrderedDict()
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = short(r['Kernel_Name'])
        a = acc.setdefault(name, collections.defaultdict(float))
        a[r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen[name]:
            seen[name].add(r['Dispatch_Id'])
            a['_ms'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    print('==', d)
    for name, a in acc.items():
        if a['_ms'] < 0.02:
            continue
        print('%-44s n=%-4d ms=%8.3f  ' % (name, len(seen[name]), a['_ms']) +
              '  '.join('%s=%.4g' % (k, v) for k, v in a.items() if k != '_ms'))
