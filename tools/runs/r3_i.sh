#!/bin/bash
# Round 3, step i: max-pool and ego-warp of the training graph on the library's kernels (both directions): GPU tests of the
# training tier, the training step's time against round 3's earlier figure (62.1 ms at B = 2 from the lifted features).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_i
mkdir -p $O
timeout 1200 python -m pytest tests/test_train_graph.py -q -m gpu -x > $O/pytest_train.txt 2>&1; tail -3 $O/pytest_train.txt
{
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --torch-conv 2>&1 | grep time_train_step
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --from-images 2>&1 | grep time_train_step
} > $O/train_step.txt
cat $O/train_step.txt
