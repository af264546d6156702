#!/bin/bash
# dense small-k passes (262 144 tips x 32 characters): the in-tree library (A) against scratch/$1 (B), twice each
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cp $R/pastml_amd/libpastml_hip.so /tmp/libA.so
for v in A B A B; do
  case $v in A) cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so;; B) cp $R/scratch/$1 $R/pastml_amd/libpastml_hip.so;; esac
  echo "== $v"; python3 $R/scripts/r04_ragged.py ${CASES:-balanced4 balanced12 ragged4 ragged12}
done
cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so
