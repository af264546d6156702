#!/bin/bash
# runs a script with two builds of the library inside one call: A = in-tree, B = scratch/$1; the rest are the script's args
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$1; shift
cd /tmp && export TMPDIR=/tmp
cp $R/pastml_amd/libpastml_hip.so /tmp/libA.so
for v in A B A2 B2; do
  case $v in A*) cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so;; B*) cp $R/scratch/$B $R/pastml_amd/libpastml_hip.so;; esac
  echo "== $v"; timeout -k 10 300 python3 "$@" 2>&1 | tail -2
done
cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so
