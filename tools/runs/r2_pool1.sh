#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2p1
mkdir -p $O
(
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pool or geometry or projection or splat or warp or host_camera" 2>&1 | tail -5
echo "== compact default"; timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact=0 (dense plane)"; FIERY_POOL_COMPACT=0 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact prepass rows 14"; FIERY_POOL_PREPASS_ROWS=14 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact prepass rows 28"; FIERY_POOL_PREPASS_ROWS=28 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact cells 37000 (one 1024-thread wg per CU)"; FIERY_POOL_CELLS=37000 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact tail parts 1"; FIERY_POOL_TAIL_PARTS=1 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
echo "== compact tail parts 4"; FIERY_POOL_TAIL_PARTS=4 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/microbench.py pool --reps 10 > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $O/kernel_stats.csv "microbench pool --reps 10, compact form" && cat $O/kernel_stats.csv | head -20
) 2>&1 | tee $O/pool1.txt
