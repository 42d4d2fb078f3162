#!/bin/bash
# One translation unit rebuilt with extra flags and linked with the in-tree objects of the other units into
# tools/ab/libfiery_hip_<name>.so (select with FIERY_HIP_LIB=...): same-box A/B of kernel variants without a full build.
#   tools/ab_variant.sh <name> <unit.hip> [flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p fiery_amd/build_ab tools/ab
extra=""
case $unit in lift_splat*.hip|warp.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ifiery_amd/csrc -x hip -DFIERY_CONV_TUNING=0 $extra "$@" \
  -c fiery_amd/csrc/$unit -o fiery_amd/build_ab/${name}_$unit.o
objs=$(ls fiery_amd/build/*.o | grep -v "/${BASE_UNIT:-$unit}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libfiery_hip_$name.so $objs fiery_amd/build_ab/${name}_$unit.o
echo tools/ab/libfiery_hip_$name.so
