#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout -k 10 600 python scripts/r05_relabel.py 262144 64 4 12
R05_ARITY=4 timeout -k 10 600 python scripts/r05_relabel.py 100000 64 4
timeout -k 10 600 python scripts/r05_relabel.py 20000 64 4
R05_COLS=14 timeout -k 10 600 python scripts/r05_relabel.py hiv1c 12 2
R05_COLS=246 timeout -k 10 600 python scripts/r05_relabel.py hiv1c 2
} 2>&1 | grep -v Warning | tee gpurun_out/r05e_relabel.txt
