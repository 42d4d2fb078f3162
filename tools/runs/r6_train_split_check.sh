#!/bin/bash
# the trainer-fixture test (forward outputs against the reference's fp32 CPU fixture) with and without the split forms in the training graph
mkdir -p gpurun_out/train_split; O=gpurun_out/train_split
for v in 0 1; do
  FIERY_TRAIN_SPLIT=$v timeout 600 python -m pytest tests/test_train_graph.py -q -m gpu -k "trainer_step_from_images or as_close_to_exact" > $O/fixture_split$v.txt 2>&1
  echo "== FIERY_TRAIN_SPLIT=$v"; grep "trainer_step_tiny\|training step\|passed\|failed" $O/fixture_split$v.txt | cut -c1-170
done
