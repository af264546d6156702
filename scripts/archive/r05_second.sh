#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "pij or eigen or jtt or custom or matrix or Pij or hky" > gpurun_out/r05b_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05b_pytest.log
for sr in 16 32 64; do
  PASTML_HIP_PIJ_STAGE_ROWS=$sr timeout -k 10 300 python scripts/r05_pij.py 20 32 17 24 2>&1 | tee -a gpurun_out/r05b_pij.txt
done
timeout -k 10 600 python scripts/r04_year_trace.py > gpurun_out/r05b_year.txt 2>&1; echo "year rc=$?"; head -8 gpurun_out/r05b_year.txt
