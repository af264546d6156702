#!/bin/bash
# per-level kernel times (rocprofv3 kernel trace) of two builds of the library inside one call: A = in-tree, B = scratch/$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) [ -f "$R/scratch/$1" ] || continue; export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  rm -rf $O/ktab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktab_$v -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-validate > /dev/null 2> $O/ktab_$v.err || exit 1
  echo "== $v"; python3 $R/scripts/kt_levels.py $O/ktab_$v | tail -14
done
unset PASTML_HIP_LIBRARY
