#!/bin/bash
# fused sweeps of 65 - 128-state eigen models on a larger tree (65 536 tips x 4 characters), then a kernel trace at k = 128
cd "$(dirname "$0")/.."
LEVELS=16 python3 scripts/r06_eigen_fused_wide.py 67 100 128 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp LEVELS=16
rm -rf /tmp/eigf && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eigf -o eigf -- python3 /root/repo/scripts/r06_eigen_fused_wide.py 128 > /tmp/eigf.log 2>&1 || { tail -5 /tmp/eigf.log; exit 1; }
python3 - <<'PY'
import csv, glob
fs = glob.glob('/tmp/eigf/**/*kernel_stats*.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
for r in rows[:12]:
    print('{:86s} calls {:>6s}  total {:>10.3f} ms  avg {:>9.4f} ms  {:>5s} %'.format(r['Name'][:86], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
PY
