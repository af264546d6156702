#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05f_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r05f_pytest.log
