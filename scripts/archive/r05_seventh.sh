#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for tdt in 1 0; do
  if [ $tdt = 1 ]; then export PASTML_HIP_NO_TD_TAIL=1; else unset PASTML_HIP_NO_TD_TAIL; fi
  echo "== NO_TD_TAIL=$tdt"
  timeout -k 10 300 python scripts/r04_ragged.py ragged4 ragged12 ragged64 poly64 balanced4 2>&1 | grep -v Warn
done
} | tee gpurun_out/r05k_td_tail.txt
unset PASTML_HIP_NO_TD_TAIL
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05k_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r05k_pytest.log
