#!/bin/bash
# Round 4: operator table of one training step (where the ATen glue is)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_o
mkdir -p $O
timeout 600 python tools/time_train_step.py --batch 2 --steps 3 --profile --ops > $O/train_ops.txt 2>&1
grep time_train_step $O/train_ops.txt | cut -c1-300
