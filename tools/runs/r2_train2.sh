#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
nproc > gpurun_out/r2_train/diag.txt
timeout 600 python tools/runs/r2_train_diag.py cuda >> gpurun_out/r2_train/diag.txt 2>&1
timeout 600 python tools/runs/r2_train_diag.py cpu >> gpurun_out/r2_train/diag.txt 2>&1
grep -v "^  File\|^$" gpurun_out/r2_train/diag.txt | tail -12
