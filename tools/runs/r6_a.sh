#!/bin/bash
# Round 6, first GPU call: the pooling op as one launch pair against frame groups on two streams and the 384-thread form.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_a}
mkdir -p $O
timeout 900 python tools/runs/r6_pool_ab.py ${VARIANTS:-"-" "FIERY_POOL_THREADS=384" "GROUPS=4+5" "GROUPS=3+6" "GROUPS=2+7" "GROUPS=3+3+3" "GROUPS=4+5,FIERY_POOL_THREADS=384" "GROUPS=3+6,FIERY_POOL_THREADS=384"} 2>&1 | grep -v amdgpu.ids | tee $O/pool_ab.txt
