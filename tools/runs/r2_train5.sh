#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_train_graph.py -m gpu -x -q -k "training_step_on_the_gpu" > gpurun_out/r2_train/pytest_only7.txt 2>&1
echo "only7 rc=$?" >> gpurun_out/r2_train/pytest_only7.txt
grep -v "^  File" gpurun_out/r2_train/pytest_only7.txt | tail -25
timeout 900 python -X faulthandler -m pytest tests/test_train_graph.py -m gpu -x -q -k "full_size_training" > gpurun_out/r2_train/pytest_only8.txt 2>&1
echo "only8 rc=$?" >> gpurun_out/r2_train/pytest_only8.txt
grep -v "^  File" gpurun_out/r2_train/pytest_only8.txt | tail -12
