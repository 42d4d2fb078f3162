#!/bin/bash
mkdir -p gpurun_out/r2_train
timeout 600 python tools/time_train_step.py --batch 2 --steps 3 --profile --ops > gpurun_out/r2_train/step_ops.txt 2>&1
grep time_train_step gpurun_out/r2_train/step_ops.txt
