#!/bin/bash
# Round 5: training step after a change to the training graph: step time, ATen kernels by autograd node, GPU training tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_w
mkdir -p $O
for rep in 1 2; do
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step | sed "s/^/rep $rep: /"
done 2>&1 | tee $O/train_step.txt
FIERY_SITES=400 timeout 500 python tools/time_train_step.py --batch 2 --steps 3 --sites > $O/sites.txt 2>&1; grep -A24 "ATen kernels" $O/sites.txt | cut -c1-200
timeout 1200 python -m pytest tests/test_train_graph.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|gradients: 322 of|Error" | tee $O/pytest.txt
