#!/bin/bash
# builds a variant of the library for a same-box A/B: scripts/r06_variant.sh <name> <unit.hip> "<extra flags>"
# (the unit is recompiled with the flags, every other object comes from pastml_amd/csrc/build) -> scratch/r06/lib_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; flags=$3
mkdir -p $R/scratch/r06
cd $R/pastml_amd/csrc
obj=$R/scratch/r06/${name}_${unit%.hip}.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC $flags -c -o $obj $unit
others=$(ls build/*.o | grep -v "build/${unit%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/r06/lib_$name.so $obj $others -ldl
echo built $R/scratch/r06/lib_$name.so
