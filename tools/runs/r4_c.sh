#!/bin/bash
# Round 4: prepass rows in flight; raw timeline dump of the pooling workgroups.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_c
mkdir -p $O
TRACE=1 TRACE_DUMP=$O ROUNDS=2 timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_TAIL_PARTS=4" "FIERY_POOL_TAIL_PARTS=4,FIERY_POOL_PREPASS_ROWS=14" "FIERY_POOL_TAIL_PARTS=4,FIERY_POOL_PREPASS_ROWS=28" "FIERY_POOL_TAIL_PARTS=4,FIERY_POOL_PREPASS_ROWS=4" 2>&1 | tee $O/pool_prepass_rows.txt
