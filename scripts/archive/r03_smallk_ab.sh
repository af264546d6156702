#!/bin/bash
# small-k many-column throughput with two builds: A = in-tree, B = scratch/$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  for k in 2 4 8 12 20; do
    timeout -k 10 120 python3 $R/scripts/smallk_rate.py 18 $k 32 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v k=$k', 'bu %.3f td %.3f prep %.3f total %.3f ms' % (d['bottom_up']['ms'], d['top_down']['ms'], d['prep']['ms'], d['total_ms']))"
  done
done
unset PASTML_HIP_LIBRARY
