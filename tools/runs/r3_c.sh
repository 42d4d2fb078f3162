#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_c
mkdir -p $O
echo "== reservations kept (addresses never reused)"
timeout 300 python tools/guard_alloc/selftest.py 2>&1 | grep -v amdgpu.ids | tee $O/guard_selftest_keep_va.txt | tail -14
echo "== reservations freed"
FIERY_GUARD_FREE_VA=1 timeout 300 python tools/guard_alloc/selftest.py 2>&1 | grep -v amdgpu.ids | tee $O/guard_selftest_free_va.txt | tail -3
