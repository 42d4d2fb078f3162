#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
POOL_TILES=20480,40000,10000 timeout 300 python tools/microbench.py pool --probe --reps 10 2>&1 | grep "pool\|probe" > $O/pool3.txt
cat $O/pool3.txt
