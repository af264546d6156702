#!/bin/bash
# what the polish / continuation of many-parameter searches costs cfg5 end to end
for env in "X=1" "PASTML_AMD_POLISH_STEP=0" "PASTML_AMD_POLISH_STEP=0 PASTML_AMD_CONTINUE=0"; do
  echo "== $env"
  env $env timeout -k 10 300 python scripts/r04_profile_acr_all.py 2>&1 | grep "acr wall" | cut -c1-160
done
