#!/usr/bin/env python3
"""Condenses the outputs of scripts/r03_profile_secondary.sh (rocprofv3 directories under argv[1]) into one markdown file."""
import collections
import csv
import glob
import os
import sys

O = sys.argv[1]


def find(tag, suffix):
    hits = glob.glob(os.path.join(O, 'r03e_' + tag, '**', '*' + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def stats(tag, title):
    path = find(tag, '_kernel_stats.csv')
    print('## {} (rocprofv3 --kernel-trace --stats)\n'.format(title))
    if not path:
        print('(missing)\n')
        return
    print('| kernel | calls | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|')
    for r in csv.DictReader(open(path)):
        if float(r['TotalDurationNs']) < 20000:
            continue
        print('| {} | {} | {:.3f} | {:.1f} | {:.1f} | {:.1f} |'.format(
            r['Name'].split('(')[0].replace('void ', ''), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
    print()


def counters(tags, title):
    acc = collections.OrderedDict()
    for tag in tags:
        path = find(tag, '_counter_collection.csv')
        if not path:
            continue
        for r in csv.DictReader(open(path)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            d = acc.setdefault(name, collections.OrderedDict())
            d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            d.setdefault('_dispatches_' + tag, set()).add(r['Dispatch_Id'])
    print('## {} (rocprofv3 --pmc, summed over the dispatches of the run)\n'.format(title))
    for name, d in acc.items():
        vals = {k: v for k, v in d.items() if not k.startswith('_')}
        if max(vals.values() or [0]) < 1e5:
            continue
        n = max(len(v) for k, v in d.items() if k.startswith('_'))
        line = ', '.join('{} {:.3g}'.format(k, v) for k, v in vals.items())
        derived = []
        if vals.get('SQ_WAVE_CYCLES'):
            if 'SQ_WAIT_ANY' in vals:
                derived.append('waiting {:.0f} % of wave cycles'.format(100 * vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES']))
            if 'SQ_ACTIVE_INST_VALU' in vals and vals.get('SQ_BUSY_CYCLES'):
                derived.append('VALU active {:.2f} of busy cycles x4'.format(vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_BUSY_CYCLES'] / 4))
        if vals.get('SQ_INSTS_VALU') and 'SQ_INSTS_MFMA' in vals:
            derived.append('MFMA share of VALU instructions {:.1f} %'.format(100 * vals['SQ_INSTS_MFMA'] / vals['SQ_INSTS_VALU']))
        if vals.get('SQ_WAVES') and vals.get('SQ_INSTS_VALU'):
            derived.append('{:.0f} VALU instructions per wave'.format(vals['SQ_INSTS_VALU'] / vals['SQ_WAVES']))
        print('* `{}` ({} dispatches): {}{}'.format(name, n, line, (' -- ' + '; '.join(derived)) if derived else ''))
    print()


print('# Kernels outside the headline: kernel stats and SQ counters (round 3, scripts/r03_profile_secondary.sh)\n')
stats('cfg3m_kt', 'cfg3 marginal pass: eigen_gemm_kernel, 262 144 tips, JTT k = 20, 10 passes')
counters(['cfg3m_sq_a', 'cfg3m_sq_b'], 'cfg3 marginal pass, 3 passes')
stats('cfg3j_kt', 'cfg3 joint sweep + back-trace: eigen_joint_kernel, 10 passes')
counters(['cfg3j_sq_a'], 'cfg3 joint sweep, 3 passes')
stats('smallk_kt', 'small k: 262 144 tips x 32 characters, k = 4, F81 level kernels, 3 marginal passes')
counters(['smallk_sq'], 'small k, 3 passes')
stats('thin_kt', 'thin levels: cfg2 marginal pass (65 536 tips, k = 4, 1 column) and the cfg5-shaped gradient (HIV1C tree, k = 12, 14 columns), 20 repetitions each')
counters(['thin_sq'], 'thin levels, 5 repetitions each')
