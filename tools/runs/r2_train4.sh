#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
FIERY_POOL_VERBOSE=1 timeout 300 python tools/runs/r2_pool_diag.py > gpurun_out/r2_train/pool_diag.txt 2>&1
echo "rc=$?" >> gpurun_out/r2_train/pool_diag.txt
FIERY_POOL_COMPACT=0 FIERY_POOL_VERBOSE=1 timeout 300 python tools/runs/r2_pool_diag.py > gpurun_out/r2_train/pool_diag_dense.txt 2>&1
echo "rc=$?" >> gpurun_out/r2_train/pool_diag_dense.txt
grep -v "^  File" gpurun_out/r2_train/pool_diag.txt | tail -15
grep -v "^  File" gpurun_out/r2_train/pool_diag_dense.txt | tail -8
