#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
{
for c in ragged4 ragged12 ragged64 poly4 balanced4 mid4; do
  timeout -k 10 300 python scripts/r05_tune_ab.py $c default= old=NARROW_UNITS:16,NO_TD_TAIL:1 notail=NO_TD_TAIL:1
done
timeout -k 10 300 python scripts/r05_tune_ab.py ragged64 default= absorb=ABSORB_MIN:64 grid16k=GRID_CAP:16384
timeout -k 10 300 python scripts/r05_tune_ab.py ragged4 default= grid16k=GRID_CAP:16384 grid64k=GRID_CAP:65536 nostage=NO_TD_STAGE:1
timeout -k 10 300 python scripts/r05_tune_ab.py ragged12 default= grid16k=GRID_CAP:16384 grid64k=GRID_CAP:65536
} 2>&1 | grep -v Warn | tee gpurun_out/r05m_ragged_switches.txt
