#!/bin/bash
# per-launch durations of one marginal pass (kernel trace only): $1 = case of r04_prof_driver.py, rest = env settings
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
CASE=$1; shift
TAG=${TAG:-$CASE}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r04lv_$TAG
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/r04lv_$TAG -o run -- python3 $R/scripts/r04_prof_driver.py $CASE 3 > $O/r04lv_$TAG.log 2>&1 || { tail -5 $O/r04lv_$TAG.log; exit 1; }
python3 - <<PY
import csv, glob
p = glob.glob('$O/r04lv_$TAG/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(p)) if 'f81' in r['Kernel_Name']]
# the last pass
starts = [i for i, r in enumerate(rows) if 'prep' in r['Kernel_Name']]
rows = rows[starts[-1]:]
t0 = int(rows[0]['Start_Timestamp'])
out = []
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    out.append('%-36s threads %9d  start %8.1f us  dur %7.1f us' % (n[:36], int(r['Grid_Size_X']) * int(r['Grid_Size_Y']), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
open('$O/r04lv_$TAG.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/r04lv_$TAG
