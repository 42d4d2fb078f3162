#!/bin/bash
# Round 4: timeline of the pooling kernel's workgroups (set-up / row stream / write-out), persistent and per-item launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_b
mkdir -p $O
TRACE=1 ROUNDS=1 timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_PERSISTENT=0" "FIERY_POOL_TAIL_PARTS=4" "-" 2>&1 | tee $O/pool_trace.txt
