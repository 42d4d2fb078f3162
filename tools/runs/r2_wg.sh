#!/bin/bash
mkdir -p gpurun_out/r2_train
python tools/runs/r2_wgrad_diag.py > gpurun_out/r2_train/wgrad_diag.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2_train/wgrad_diag.txt | tail -14
