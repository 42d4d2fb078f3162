#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
for o in 1 2; do for nt in 1 0; do
  echo "== ORDER=$o NT=$nt"
  FIERY_POOL_ORDER=$o FIERY_POOL_NT=$nt POOL_TILES=20480,40000,13334,10000 timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool"
done; done > $O/pool2.txt 2>&1
cat $O/pool2.txt
