#!/usr/bin/env python3
"""
Condenses rocprofv3 output directories (gpurun_out/...) into the summaries committed under profiles/.

    python3 profiles/summarize_rocprof.py r01 gpurun_out/prof_kt gpurun_out/pmc_fetch gpurun_out/pmc_write [workload]

* <tag>_kernel_stats.csv : rocprofv3 --kernel-trace --stats summary, verbatim
* <tag>_pmc_traffic.md   : per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs).
  Units and gfx950 correction as MI355X_MICROARCH.md section HBM prescribes: the counters are in KB; FETCH_SIZE
  reports half of the bytes of wide (16 B/lane) coalesced streaming reads, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
* traffic.json           : per-step traffic of the sweeps' kernels, read by bench.py for roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def pmc(directory, counter):
    path = max(glob.glob(os.path.join(directory, '**', '*_counter_collection.csv'), recursive=True), key=os.path.getmtime)
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc.setdefault(name, []).append((float(r['Counter_Value']),
                                         (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6))
    return acc


def main():
    tag, kt, pf, pw = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else 'cfg4'
    stats = max(glob.glob(os.path.join(kt, '**', '*_kernel_stats.csv'), recursive=True), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(HERE, tag + '_kernel_stats.csv'))
    fetch, write = pmc(pf, 'FETCH_SIZE'), pmc(pw, 'WRITE_SIZE')
    lines = ['# HBM traffic per kernel ({}; workload {}; one bench step, --pmc FETCH_SIZE and --pmc WRITE_SIZE in '
             'separate runs)'.format(tag, workload), '',
             'bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction for 16 B/lane streaming reads)', '',
             '| kernel | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | corrected GB | GB per launch | '
             'kernel ms (pmc run) | TB/s |', '|---|---|---|---|---|---|---|---|']
    traffic = {}
    for name in fetch:
        if name not in write:
            continue
        f = sum(v for v, _ in fetch[name])
        w = sum(v for v, _ in write[name])
        n = len(fetch[name])
        ms = sum(t for _, t in fetch[name])
        gb = (2 * f + w) * 1024 / 1e9
        lines.append('| {} | {} | {:.0f} | {:.0f} | {:.3f} | {:.3f} | {:.3f} | {:.2f} |'
                     .format(name, n, f, w, gb, gb / n, ms, gb / ms if ms else 0))
        traffic[name] = dict(launches=n, bytes_per_launch=(2 * f + w) * 1024 / n, fetch_kb_raw=f, write_kb=w)
    open(os.path.join(HERE, tag + '_pmc_traffic.md'), 'w').write('\n'.join(lines) + '\n')
    tj = os.path.join(HERE, 'traffic.json')
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    # bytes of one bench step over all launches of the sweep's level kernel (every template variant; not the single
    # launch that walks the narrow end, which bench.py's HIP-event bracket leaves out as well); bench.py divides by
    # the number of launches it timed
    def total(prefixes):
        sel = [v for k, v in traffic.items() if k.startswith(prefixes)]
        return sum(v['bytes_per_launch'] * v['launches'] for v in sel) if sel else None
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    # the kernel sources these counters were measured on: bench.py reports roofline.traffic only while they are unchanged
    data[workload] = dict(tag=tag, csrc_sha=bench.csrc_digest(), chars_per_gpu=bench.WORKLOADS[workload][2],
                          td_bytes_per_step=total(('td_f81_kernel', 'td_matrix_kernel', 'td_f81_super_kernel', 'td_f81_stack_kernel')),
                          bu_bytes_per_step=total(('bu_f81_kernel', 'bu_matrix_kernel', 'bu_f81_super_kernel', 'bu_f81_stack_kernel')),
                          td_two_level_bytes_per_step=total(('td_f81_super_kernel',)),
                          bu_two_level_bytes_per_step=total(('bu_f81_super_kernel',)),
                          kernels=traffic)
    json.dump(data, open(tj, 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
