#!/bin/bash
# Round 5: Winograd kernel variants, same box: per-shape launch times (tools/runs/r5_wino_times.py) for each library in LIBS
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r5_f}
mkdir -p $O
timeout 60 python tools/runs/r5_wino_times.py > /dev/null 2>&1     # throw-away: clocks settle
for rep in 1 2; do
for lib in $LIBS; do
  var=""; [ "$lib" != "-" ] && var="FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so"
  echo "== $lib (rep $rep)"
  env $var timeout 120 python tools/runs/r5_wino_times.py 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $O/summary.txt
