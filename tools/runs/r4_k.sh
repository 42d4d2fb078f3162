#!/bin/bash
# Round 4: timeline of the conv kernel's workgroups (tuning build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_k
mkdir -p $O
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_tuning.so timeout 300 python tools/runs/r4_conv_trace.py $O 2>&1 | tee $O/conv_trace.txt
