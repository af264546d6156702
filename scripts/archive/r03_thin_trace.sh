#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg5; do
rm -rf $O/thin_$w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/thin_$w -o run -- python3 $R/scripts/r03_thin_trace.py $w 20 > /dev/null 2> $O/thin_$w.err
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob('$O/thin_$w/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last repetition: from the last per-branch pass (or single-launch kernel) on
starts = [i for i, r in enumerate(rows) if 'prep' in r['Kernel_Name']]
rows = rows[starts[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
print('== $w')
for r in rows:
    print('%-44s start %8.1f us  dur %7.1f us  grid %s wg %s' % (r['Kernel_Name'].split('(')[0].replace('void ', '')[:44],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '?'), r.get('Workgroup_Size_X', '?')))
PY
done
