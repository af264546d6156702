#!/bin/bash
# A/B of two builds of the library inside one call on ONE command: pastml_amd/libpastml_hip.so (A) against scratch/$1 (B); rest = the command
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/scratch/$1
shift
cd /tmp && export TMPDIR=/tmp
cp $R/pastml_amd/libpastml_hip.so /tmp/libA.so
for v in A B A2 B2; do
  case $v in A*) cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so;; B*) cp $B $R/pastml_amd/libpastml_hip.so;; esac
  echo "== $v"; "$@" 2>&1 | tail -2
done
cp /tmp/libA.so $R/pastml_amd/libpastml_hip.so
