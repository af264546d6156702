#!/bin/bash
# the secondary configs (cfg2, cfg3, cfg5-shaped gradient) with two builds of the library: A = in-tree, B = scratch/$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) [ -f "$R/scratch/$1" ] || continue; export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  timeout -k 10 300 python3 -c "
import sys, json; sys.path.insert(0, '$R')
import bench
o = bench.secondary_measurements(0)
print('$v', {k: round(v.get('ms_per_pass') or v.get('ms_per_gradient') or v.get('seconds') or v.get('ms_per_step'), 4) for k, v in o.items()}, 'cfg3 sweep', round(o['cfg3']['ms_joint_sweep'], 4), 'marg', round(o['cfg3']['ms_marginal_pass'], 4))
" || exit 1
done
unset PASTML_HIP_LIBRARY
