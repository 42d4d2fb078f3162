#!/bin/bash
# Round 5: the GPU training-step test five times in a row (deterministic yardstick: the all-torch fp32 graph evaluated on the host
# with a fixed thread count; asserted bound 3x its distance to fp64) - the ledger row must read the same every time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_train_x5
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_train_graph.py -q -m gpu -k "training_step_on_the_gpu_is_as_close" 2>&1 | grep -E "gradients:|passed|failed" | cut -c1-260 | sed "s/^/run $i: /"
done 2>&1 | tee $O/pytest_gpu_train_x5.txt
