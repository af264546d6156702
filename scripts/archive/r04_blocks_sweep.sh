#!/bin/bash
# large ragged forests under the subtree-block schedule (normally capped at 131 072 stored nodes / 160 000 node-columns)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CASES=${CASES:-ragged64}
run() { echo "-- $*"; env "$@" timeout -k 10 120 python3 $R/scripts/r04_ragged.py $CASES 2>&1 | tail -n +1; }
run X=1
B="PASTML_HIP_BLOCK_MAX_WORK=1000000000 PASTML_HIP_BLOCK_MAX_STORED=4000000"
run $B
run $B PASTML_HIP_BLOCK_NODES=64
run $B PASTML_HIP_BLOCK_NODES=128
run $B PASTML_HIP_BLOCK_NODES=512
run $B PASTML_HIP_BLOCK_NODES=1024
run $B PASTML_HIP_BLOCK_THREADS=64
run $B PASTML_HIP_BLOCK_THREADS=128
run X=1
