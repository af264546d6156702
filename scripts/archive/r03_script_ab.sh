#!/bin/bash
# runs a script with two builds of the library inside one call: A = in-tree, B = scratch/$1; the rest are the script's args
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$1; shift
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$R/scratch/$B;; esac
  echo "== $v"; timeout -k 10 300 python3 "$@" 2>&1 | tail -2
done
unset PASTML_HIP_LIBRARY
