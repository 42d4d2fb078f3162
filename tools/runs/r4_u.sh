#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_u
mkdir -p $O
FIERY_TEST_VERBOSE=1 timeout 1200 python -m pytest tests/test_train_graph.py -q -m gpu -s -k "trainer_step or as_close_to_exact or tiny_model" 2>&1 | grep -v "amdgpu.ids" > $O/train_gpu.txt; grep -n "passed\|failed\|Error\|assert" $O/train_gpu.txt | tail -12; grep "trainer_step_tiny\|train_step\[" $O/train_gpu.txt | tail -24 | cut -c1-260
