#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_p
mkdir -p $O
timeout 600 python tools/time_train_step.py --batch 2 --steps 3 --sites > $O/train_sites.txt 2>&1
grep -v "^W2026\|amdgpu.ids" $O/train_sites.txt | tail -60 | cut -c1-220
