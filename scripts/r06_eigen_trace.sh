#!/bin/bash
# kernel trace of the fused eigen marginal pass: scripts/r06_eigen_trace.sh "<levels k columns>" <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/eig_kt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/eig_kt -o run -- python3 $R/scripts/r06_eigen_k61.py $1 0 > /dev/null 2> $R/gpurun_out/eig_kt.err
python3 - <<PY
import csv, glob
path = glob.glob('$R/gpurun_out/eig_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last marginal pass: from the last eigen_gemm_kernel<.., 1> (tips) on, until the next tips launch
idx = [i for i, r in enumerate(rows) if 'eigen_gemm_kernel' in r['Kernel_Name'] and ', 1>' in r['Kernel_Name']]
# marginal passes come first (10 + 1), then bottom-up sweeps: take the 5th pass
a = idx[4]; b = idx[5]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('%-46s grid %9s  start %8.1f us  %8.1f us' % (r['Kernel_Name'].split('(')[0].replace('void ', '')[:46], r['Grid_Size'], (int(r['Start_Timestamp']) - t0) / 1e3, us))
PY
