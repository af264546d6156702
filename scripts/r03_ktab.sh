#!/bin/bash
# per-level kernel times (rocprofv3 kernel trace) of two builds of the library inside one call: A = in-tree, B = scratch/$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cp $R/pastml_amd/libpastml_hip.so /tmp/libA.so
for v in A B A2 B2; do
  case $v in A*) cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so;; B*) [ -f "$R/scratch/$1" ] || continue; cp $R/scratch/$1 $R/pastml_amd/libpastml_hip.so;; esac
  rm -rf $O/ktab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktab_$v -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-validate > /dev/null 2> $O/ktab_$v.err || exit 1
  echo "== $v"; python3 $R/scripts/kt_levels.py $O/ktab_$v | tail -14
done
cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so
