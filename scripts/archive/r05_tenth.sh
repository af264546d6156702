#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -m gpu -x -q -k "pij or eigen or jtt or custom or matrix or Pij or hky or counts or fuzz or height" > gpurun_out/r05o_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05o_pytest.log
{
for v in 0 1; do
  if [ $v = 1 ]; then export PASTML_HIP_NO_PIJ_VALU=1; else unset PASTML_HIP_NO_PIJ_VALU; fi
  echo "== NO_PIJ_VALU=$v"
  timeout -k 10 300 python scripts/r05_pij.py 20 32 17 24 8 5
done
unset PASTML_HIP_NO_PIJ_VALU
for nb in 1024 2048 8192 16384; do
  PASTML_HIP_PIJ_BLOCKS=$nb timeout -k 10 300 python scripts/r05_pij.py 20 32 2>&1 | sed "s/^/blocks=$nb /"
done
} 2>&1 | tee gpurun_out/r05o_pij_valu.txt
