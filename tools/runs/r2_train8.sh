#!/bin/bash
mkdir -p gpurun_out/r2_train
timeout 900 python -m pytest tests/test_train_graph.py -m gpu -x -q -k "fixture" > gpurun_out/r2_train/pytest_fixture.txt 2>&1
grep -E "^E  |passed|failed|train_model_tiny" gpurun_out/r2_train/pytest_fixture.txt | cut -c1-220 | tail -14
