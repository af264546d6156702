#!/usr/bin/env python3
"""Counters of the largest eigen_fused launch: scripts/pmc_cfg3_big.py gpurun_out/TAG"""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
KN = sys.argv[2] if len(sys.argv) > 2 else 'eigen_fused'
out = {}
kt = glob.glob(tag + '_kt/**/*kernel_trace.csv', recursive=True)
if kt:
    rows = [r for r in csv.DictReader(open(kt[0])) if KN in r['Kernel_Name']]
    durs = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    out['duration_us_max'] = durs[-1] / 1e3
    out['duration_us_all'] = sum(durs) / 1e3 / 4  # 4 sweeps (1 warm-up + 3)
for part in 'abc':
    f = glob.glob(tag + '_' + part + '/**/*counter_collection.csv', recursive=True)
    if not f:
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if KN not in r['Kernel_Name']:
            continue
        d = disp.setdefault(int(r['Dispatch_Id']), {'grid': int(r['Grid_Size'])})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    # the largest level: most waves / most VALU; take the dispatch with the largest first counter value
    best = max(disp.values(), key=lambda d: max(v for k, v in d.items() if k != 'grid'))
    out.update({k: v for k, v in best.items() if k != 'grid'})
    out['grid_' + part] = best['grid']
for k, v in out.items():
    print('%-28s %14.4g' % (k, v))
if out.get('SQ_INSTS_MFMA'):
    print('VALU per MFMA %.1f, SALU per MFMA %.1f, LDS per MFMA %.2f' % (
        out['SQ_INSTS_VALU'] / out['SQ_INSTS_MFMA'], out['SQ_INSTS_SALU'] / out['SQ_INSTS_MFMA'],
        out['SQ_INSTS_LDS'] / out['SQ_INSTS_MFMA']))

if 'SQ_ACTIVE_INST_VALU' in out and 'SQ_BUSY_CYCLES' in out:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs (1024); SQ_BUSY_CYCLES cycles summed over 32 SEs
    cycles = out['SQ_BUSY_CYCLES'] / 32
    print('kernel cycles %.0f (%.2f GHz); VALU busy %.3f of the SIMD cycles; MFMA busy %.3f; waves per SIMD %.2f' % (
        cycles, cycles / out['duration_us_max'] / 1e3, out['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cycles,
        out.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cycles, out['SQ_WAVE_CYCLES'] * 4 / 1024 / cycles))
