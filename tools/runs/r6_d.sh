#!/bin/bash
# Round 6: the pooling op of this tree against round 5's translation unit (tools/ab/libfiery_hip_r5pool.so), same box, alternating.
# Each process runs its variants twice and the second round counts (the first rows of a fresh process read low: clocks settle).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_d}
mkdir -p $O
for rep in 1 2; do
  echo "== round 5 translation unit (rep $rep)"
  FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_r5pool.so ROUNDS=2 timeout 300 python tools/runs/r6_pool_ab.py "-" 2>&1 | grep "^\[1"
  echo "== this tree (rep $rep)"
  ROUNDS=2 timeout 300 python tools/runs/r6_pool_ab.py ${VARIANTS:-"-" "NO_RANKS=1"} 2>&1 | grep "^\[1"
done | tee $O/pool_r5_vs_r6.txt
