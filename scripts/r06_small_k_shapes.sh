#!/bin/bash
# k <= 8 on large forests: states per lane of the F81 level kernels (default 2; F81_R / F81_TD_R = 4: one / two lanes per unit)
cd "$(dirname "$0")/.."
for c in ragged4 ragged8; do
  for rep in 1 2; do
    python3 scripts/tune_one.py $c default= 2>&1 | grep -v amdgpu.ids
    python3 scripts/tune_one.py $c bu4=F81_R:4 2>&1 | grep -v amdgpu.ids
    python3 scripts/tune_one.py $c td4=F81_TD_R:4 2>&1 | grep -v amdgpu.ids
    python3 scripts/tune_one.py $c both4=F81_R:4,F81_TD_R:4 2>&1 | grep -v amdgpu.ids
  done
done
