#!/bin/bash
# Round 5: Winograd in the training graph's 3x3 convolutions (forward + input gradient): step time off / on, then the GPU training tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_v
mkdir -p $O
for rep in 1 2; do for w in 0 1; do
  FIERY_TRAIN_WINOGRAD=$w timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step | sed "s/^/FIERY_TRAIN_WINOGRAD=$w rep $rep: /"
done; done 2>&1 | tee $O/train_step.txt
timeout 1200 python -m pytest tests/test_train_graph.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|gradients: 322 of" | tee $O/pytest.txt
