#!/bin/bash
# A/B of two builds of the library inside one call on ONE command: pastml_amd/libpastml_hip.so (A) against scratch/$1 (B); rest = the command
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/scratch/$1
shift
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A2 B2; do
  case $v in A*) unset PASTML_HIP_LIBRARY;; B*) export PASTML_HIP_LIBRARY=$B;; esac
  echo "== $v"; "$@" 2>&1 | tail -2
done
unset PASTML_HIP_LIBRARY
