#!/bin/bash
# per-launch durations of one marginal pass, thin ends on and off
set -e
TAG=ragged4_thin bash scripts/r04_levels_trace.sh ragged4 PASTML_HIP_DEBUG=1 > /dev/null
TAG=ragged4_nothin bash scripts/r04_levels_trace.sh ragged4 PASTML_HIP_NO_THIN=1 > /dev/null
TAG=ragged12_thin bash scripts/r04_levels_trace.sh ragged12 PASTML_HIP_DEBUG=1 > /dev/null
grep pastml_hip gpurun_out/r04lv_ragged4_thin.log || true
