#!/bin/bash
# Round 4: phases of the conv kernel's workgroups (tuning build: set-up / K loop / epilogue cycles of sampled workgroups)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_i
mkdir -p $O
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_tuning.so timeout 300 python tools/microbench.py conv --reps 20 --clk 2>&1 | tee $O/conv_clk.txt
