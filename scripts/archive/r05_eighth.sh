#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for c in ragged4 ragged12 ragged64 poly4 balanced4 mid4; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $c old=NARROW_UNITS:16,NO_TD_TAIL:1 floor=NO_TD_TAIL:1 floor_tail= n64=NARROW_UNITS:64 n256=NARROW_UNITS:256 2>&1 | grep -v Warn
done | tee gpurun_out/r05l_narrow_ab.txt
