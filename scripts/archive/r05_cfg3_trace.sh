#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r05_cfg3_kt
PASTML_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/r05_cfg3_kt -o run -- python3 $R/scripts/cfg3_run.py > $O/r05_cfg3_kt.log 2>&1 || { tail -5 $O/r05_cfg3_kt.log; exit 1; }
python3 - <<PY
import csv, glob
p = glob.glob('$O/r05_cfg3_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(p))]
# last joint pass: from the last obs_tips kernel on
idx = [i for i, r in enumerate(rows) if 'obs_tips' in r['Kernel_Name']]
rows = rows[idx[-1] - 1:]
t0 = int(rows[0]['Start_Timestamp'])
out = []
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    out.append('%-40s grid %7d x %3d  start %8.1f us  dur %7.1f us' % (n[:40], int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Y']), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
open('$O/r05_cfg3_trace.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/r05_cfg3_kt
