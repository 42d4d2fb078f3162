#!/bin/bash
# Round 4: the conv workgroups' timeline with the epilogue at wave priority 3 against the default (tuning builds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_s
mkdir -p $O
for v in tuning tuning_prio3; do
  echo "== $v"
  FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$v.so timeout 300 python tools/runs/r4_conv_trace.py 2>&1 | grep -v amdgpu.ids | grep -A2 "persistent=0"
done | tee $O/conv_trace_prio.txt
