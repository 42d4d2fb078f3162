#!/bin/bash
# training tests on the GPU + step timing with the operator profile
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_graph.py -m gpu -x -q > gpurun_out/r2_train/pytest_train.txt 2>&1
echo "pytest train rc=$?" >> gpurun_out/r2_train/pytest_train.txt
grep -E "^E  |passed|failed|rc=" gpurun_out/r2_train/pytest_train.txt | cut -c1-200 | tail -6
timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --profile > gpurun_out/r2_train/step_hip.txt 2>&1
grep "k_conv_wgrad\|time_train_step" gpurun_out/r2_train/step_hip.txt | cut -c1-50,140-330
FIERY_WGRAD_STAGED=0 timeout 600 python tools/time_train_step.py --batch 2 --steps 5 > gpurun_out/r2_train/step_hip_unstaged.txt 2>&1
grep "time_train_step" gpurun_out/r2_train/step_hip_unstaged.txt
