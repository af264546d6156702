#!/bin/bash
# kernel durations of scripts/r04_thin.py: the in-tree library (A) against scratch/$1 (B), rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B; do
  case $v in A) unset PASTML_HIP_LIBRARY;; B) export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  rm -rf $O/r04thin_$v
  rocprofv3 --kernel-trace --output-format csv -d $O/r04thin_$v -o run -- python3 $R/scripts/r04_thin.py > $O/r04thin_$v.log 2>&1 || { tail -5 $O/r04thin_$v.log; exit 1; }
  echo "== $v"; grep "^hiv1c\|^cfg2" $O/r04thin_$v.log
  python3 - <<PY
import csv, glob, collections
t = glob.glob('$O/r04thin_$v/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
seq = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'small' in n or 'blocks' in n:
        seq.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, d in seq.items():
    print('%-34s' % n[:34], ' '.join('%7.1f' % (sum(d[i:i + 100]) / len(d[i:i + 100])) for i in range(0, len(d), 100)), ' us (per 100 launches)')
PY
  rm -rf $O/r04thin_$v
done
unset PASTML_HIP_LIBRARY
if [ -n "$DENSE" ]; then
  for v in A B; do
    case $v in A) unset PASTML_HIP_LIBRARY;; B) export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
    echo "== $v dense levels"; python3 $R/scripts/r04_ragged.py balanced4 balanced12 ragged4 ragged12
  done
  unset PASTML_HIP_LIBRARY
fi
