#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_graph.py -m gpu -x -q > gpurun_out/r2_train/pytest_train.txt 2>&1
echo "pytest train rc=$?" >> gpurun_out/r2_train/pytest_train.txt
tail -4 gpurun_out/r2_train/pytest_train.txt
timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --profile --ops > gpurun_out/r2_train/step_hip.txt 2>&1
grep time_train_step gpurun_out/r2_train/step_hip.txt
