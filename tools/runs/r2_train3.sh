#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_graph.py -m gpu -x -q > gpurun_out/r2_train/pytest_train.txt 2>&1
echo "pytest train rc=$?" >> gpurun_out/r2_train/pytest_train.txt
grep -v "^  File" gpurun_out/r2_train/pytest_train.txt | tail -30
for w in 2048 4096 8192; do
  FIERY_WGRAD_WGS=$w timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --profile > gpurun_out/r2_train/step_hip_w$w.txt 2>&1
  echo "wgs=$w"; grep "k_conv_wgrad\|time_train_step" gpurun_out/r2_train/step_hip_w$w.txt | cut -c1-40,140-400
done
