#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ab in 0 1 2; do
  PASTML_HIP_PIJ_ABLATE=$ab timeout -k 10 300 python scripts/r05_pij.py 20 2>&1 | sed "s/^/ablate=$ab /" | tee -a gpurun_out/r05c_pij_ablate.txt
done
for combo in "0 0" "1 0" "0 1e-6" "1 1e-6" "0 1e-5" "0 1e-7"; do
  set -- $combo
  PASTML_AMD_CONTINUE=$1 PASTML_AMD_POLISH_STEP=$2 timeout -k 10 300 python scripts/r05_year.py 2>&1 | grep -v Warning | tee -a gpurun_out/r05c_year.txt
done
