#!/bin/bash
# the training step with the split forms (Winograd forward / input gradient, 3 x 3 weight gradient) beside FIERY_TRAIN_SPLIT=0; GPU training tests
mkdir -p gpurun_out/train_split; O=gpurun_out/train_split
timeout 1500 python -m pytest tests/test_train_graph.py tests/test_reference_callers.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for r in 1 2; do for v in 1 0; do
  echo -n "FIERY_TRAIN_SPLIT=$v  "; FIERY_TRAIN_SPLIT=$v timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step | cut -c1-260
done; done | tee $O/times.txt
