#!/bin/bash
# kernel trace of the fused marginal pass of an eigen model with K states (default 128): scripts/r06_eigen_fused_wide.py K under rocprofv3
K=${1:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/eigf && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eigf -o eigf -- python3 /root/repo/scripts/r06_eigen_fused_wide.py $K > /tmp/eigf.log 2>&1 || { tail -5 /tmp/eigf.log; exit 1; }
grep -v amdgpu.ids /tmp/eigf.log
python3 - <<'PY'
import csv, glob
fs = glob.glob('/tmp/eigf/**/*kernel_stats*.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
for r in rows[:14]:
    print('{:86s} calls {:>6s}  total {:>10.3f} ms  avg {:>9.4f} ms  {:>5s} %'.format(r['Name'][:86], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
PY
