#!/bin/bash
# Round 4: round 3's pooling translation unit (linked into today's library) against today's, on one box, alternating processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_aa
mkdir -p $O
for rep in 1 2; do
  echo "== round 3 lift_splat.hip (rep $rep)"
  FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_r3pool.so ROUNDS=2 REPS=40 timeout 300 python tools/runs/r4_pool_ab.py "-" 2>&1 | grep "^\["
  echo "== today's (rep $rep)"
  ROUNDS=2 REPS=40 timeout 300 python tools/runs/r4_pool_ab.py "-" "PY_CLEAN=1" "PY_CLEAN=1,FIERY_POOL_PART_RANGES=1" 2>&1 | grep "^\["
done | tee $O/pool_r3_vs_r4.txt
