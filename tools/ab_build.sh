#!/bin/bash
# Builds the library of another git revision beside the working tree's, for A/B runs on one GPU box:
#   tools/ab_build.sh <rev> <name>   ->  tools/ab/libfiery_hip_<name>.so   (use with FIERY_HIP_LIB=...)
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" fiery_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/tools/ab"
objs=""
for f in runtime.cpp lift_splat.hip warp.hip conv_igemm.hip aux_ops.hip; do
  extra=""; [ "$f" = lift_splat.hip ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$tmp/include" -I"$tmp/fiery_amd/csrc" -x hip $extra -c "$tmp/fiery_amd/csrc/$f" -o "$tmp/$f.o"
  objs="$objs $tmp/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab/libfiery_hip_$name.so" $objs
rm -rf "$tmp"
echo "$root/tools/ab/libfiery_hip_$name.so"
