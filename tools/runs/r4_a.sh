#!/bin/bash
# Round 4, first call: pooling with persistent workgroups (tail parts taken by the same workgroups) against round 3's launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_a
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "voxel_pool or pooling" 2>&1 | tail -3
timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_PERSISTENT=0" "-" "FIERY_POOL_TAIL_PARTS=4" "FIERY_POOL_PERSISTENT=0,FIERY_POOL_TAIL_PARTS=8" "FIERY_POOL_NT=0" 2>&1 | tee $O/pool_ab.txt
timeout 300 python tools/runs/r4_pool_ab.py literature/pon_setting.yml "FIERY_POOL_PERSISTENT=0" "-" 2>&1 | tee $O/pool_ab_pon.txt
