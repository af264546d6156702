#!/bin/bash
# eigen models beyond 64 states (materialised P(t)): where the time goes.  balanced 16 384-tip tree x 4 characters
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for k in 67 100 128; do
  python3 scripts/r06_eigen_k61.py 14 $k 4 0 2>/dev/null | python3 -c "
import json,sys
r=json.load(sys.stdin)
print('k = $k  marginal pass {:.2f} ms  bottom-up {:.2f} ms'.format(r['ms_marginal_pass'], r['ms_bottom_up_sweep']))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/eigw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eigw -o eigw -- python3 /root/repo/scripts/r06_eigen_k61.py 14 128 4 0 > /tmp/eigw.log 2>&1 || { tail -5 /tmp/eigw.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/eigw/**/*kernel_stats*.csv', recursive=True)
if not fs:
    print(glob.glob('/tmp/eigw/**', recursive=True)); raise SystemExit(1)
f = fs[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print('{:70s} calls {:>6s}  total {:>10.3f} ms  avg {:>9.3f} ms  {:>5s} %'.format(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage'][:5]))
PY
