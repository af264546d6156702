#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in 1 7 5 8 1 7 5 8; do echo "-- PASTML_HIP_DBG=$d"; PASTML_HIP_DBG=$d python3 $R/scripts/r04_ragged.py ragged64; done
TAG=dbg8 bash $R/scripts/r04_levels_trace.sh ragged64 PASTML_HIP_DBG=8 | head -12
