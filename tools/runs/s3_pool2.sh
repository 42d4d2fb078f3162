#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s3b
mkdir -p $O
for t in 20480 40000 13334; do
for b in 7 8 14 16; do
  echo "tile=$t FIERY_POOL_BATCH=$b"; POOL_TILES=$t FIERY_POOL_BATCH=$b timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool frames=9 tile\|cold"
done; done 2>&1 | tee $O/pool_sweep2.txt
