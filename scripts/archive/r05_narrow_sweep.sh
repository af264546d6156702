#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for nu in default 32 64 128 256; do
  if [ $nu = default ]; then unset PASTML_HIP_NARROW_UNITS; else export PASTML_HIP_NARROW_UNITS=$nu; fi
  echo "== NARROW_UNITS=$nu"
  timeout -k 10 300 python scripts/r04_ragged.py ragged4 ragged12 ragged64 balanced4 2>&1 | grep -v Warn
done | tee gpurun_out/r05j_narrow_units.txt
