#!/bin/bash
# Round 4: issue priority for the later workgroup of every CU (does it close the 12 % gap between a CU's two workgroups?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_f
mkdir -p $O
TRACE=1 TRACE_DUMP=$O ROUNDS=2 REPS=40 timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_PERSISTENT=0" "FIERY_POOL_PERSISTENT=0,FIERY_POOL_LATE_PRIO=1" "FIERY_POOL_PERSISTENT=0,FIERY_POOL_LATE_PRIO=3" "FIERY_POOL_TAIL_PARTS=8,FIERY_POOL_LATE_PRIO=1" "FIERY_POOL_TAIL_PARTS=8,FIERY_POOL_LATE_PRIO=3" 2>&1 | tee $O/pool_ab.txt
