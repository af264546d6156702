#!/bin/bash
# the ragged 262 144-tip pass under the library's tuning switches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CASES=${CASES:-ragged64}
run() { echo "-- $*"; env "$@" python3 $R/scripts/r04_ragged.py $CASES 2>&1 | tail -n +1; }
run X=1
run PASTML_HIP_F81_R=4
run PASTML_HIP_F81_R=8
run PASTML_HIP_F81_TD_R=4
run PASTML_HIP_GRID_CAP=32768
run PASTML_HIP_GRID_CAP=4096
run PASTML_HIP_GRID_CAP=2048
run PASTML_HIP_NO_SHAPE_SORT=1
run X=1
