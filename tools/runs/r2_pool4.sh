#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2p4
mkdir -p $O
(
for A in 0 15; do
echo "== ablate $A (parts 4)"; FIERY_POOL_TAIL_PARTS=4 FIERY_POOL_ABLATE=$A timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9 tile"
done
echo "== ablate 15, parts 1"; FIERY_POOL_TAIL_PARTS=1 FIERY_POOL_ABLATE=15 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9 tile"
echo "== ablate 15, parts 8"; FIERY_POOL_TAIL_PARTS=8 FIERY_POOL_ABLATE=15 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9 tile"
echo "== frames 8 (no tail), ablate 0"; POOL_FRAMES=8 FIERY_POOL_ABLATE=0 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames"
echo "== frames 8 (no tail), ablate 15"; POOL_FRAMES=8 FIERY_POOL_ABLATE=15 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames"
) 2>&1 | tee $O/pool4.txt
