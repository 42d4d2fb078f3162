#!/bin/bash
# Round 6: the pooling op's bracket behind different predecessors (GPU-side sleep, matmuls, the geometry kernel, another pooling op)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_c}
mkdir -p $O
PRE_MODES=1 ROUNDS=2 timeout 600 python tools/runs/r6_pool_ab.py "-" "GROUPS=4+5" 2>&1 | grep -v amdgpu.ids | tee $O/pre_modes.txt
