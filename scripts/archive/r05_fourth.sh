#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "pij or eigen or jtt or custom or matrix or Pij or hky" > gpurun_out/r05d_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05d_pytest.log
for nb in 512 1024 2048 4096; do
  PASTML_HIP_PIJ_BLOCKS=$nb timeout -k 10 300 python scripts/r05_pij.py 20 32 2>&1 | sed "s/^/blocks=$nb /" | tee -a gpurun_out/r05d_pij.txt
done
for ab in 1 2; do
  PASTML_HIP_PIJ_ABLATE=$ab timeout -k 10 300 python scripts/r05_pij.py 20 2>&1 | sed "s/^/ablate=$ab /" | tee -a gpurun_out/r05d_pij.txt
done
for sr in 16 64; do
  PASTML_HIP_PIJ_STAGE_ROWS=$sr timeout -k 10 300 python scripts/r05_pij.py 20 32 17 24 2>&1 | tee -a gpurun_out/r05d_pij.txt
done
for combo in "2 1e-6" "1 1e-6"; do
  set -- $combo
  PASTML_AMD_CONTINUE=$1 PASTML_AMD_POLISH_STEP=$2 timeout -k 10 300 python scripts/r05_year.py 2>&1 | grep -v Warning | tee -a gpurun_out/r05d_year.txt
done
for combo in "0 0" "2 1e-6" "1 1e-6"; do
  set -- $combo
  PASTML_AMD_CONTINUE=$1 PASTML_AMD_POLISH_STEP=$2 timeout -k 10 300 python scripts/r05_year.py all 2>&1 | grep -v Warning | grep "columns in\|worst\|Year" | tee -a gpurun_out/r05d_year.txt
  PASTML_AMD_CONTINUE=$1 PASTML_AMD_POLISH_STEP=$2 timeout -k 10 300 python scripts/r05_year.py all 2>&1 | grep -v Warning | grep "columns in" | tee -a gpurun_out/r05d_year.txt
done
timeout -k 10 600 python scripts/r05_relabel.py 262144 64 4 2>&1 | tee gpurun_out/r05d_relabel.txt
