#!/bin/bash
# dense small-k passes (262 144 tips x 32 characters): the in-tree library (A) against scratch/$1 (B), twice each
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
# (build B is selected through PASTML_HIP_LIBRARY: the in-tree library is never overwritten)
for v in A B A B; do
  case $v in A) unset PASTML_HIP_LIBRARY;; B) export PASTML_HIP_LIBRARY=$R/scratch/$1;; esac
  echo "== $v"; python3 $R/scripts/r04_ragged.py ${CASES:-balanced4 balanced12 ragged4 ragged12}
done
unset PASTML_HIP_LIBRARY
