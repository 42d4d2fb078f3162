#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_w
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "voxel_pool or pooling" 2>&1 | tail -2
TRACE=1 ROUNDS=3 REPS=40 timeout 600 python tools/runs/r4_pool_ab.py "PY_CLEAN=1" "PY_CLEAN=1,FIERY_POOL_TAIL_PARTS=8" "PY_CLEAN=1,FIERY_POOL_TAIL_PARTS=6" 2>&1 | grep -v amdgpu.ids | tee $O/pool_ab.txt
