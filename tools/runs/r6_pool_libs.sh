#!/bin/bash
# Round 6: pooling op, several library builds on one box, alternating (LIBS="r5pool nohand -"; "-" = the in-tree library), variants in VARIANTS
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_e}
mkdir -p $O
for rep in 1 2; do
for lib in $LIBS; do
  var=""; [ "$lib" != "-" ] && var="FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so"
  echo "== $lib (rep $rep)"
  env $var ROUNDS=2 timeout 300 python tools/runs/r6_pool_ab.py ${VARIANTS:-"-"} 2>&1 | grep "^\[1"
done; done | tee $O/summary.txt
