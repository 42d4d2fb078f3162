#!/bin/bash
# Round 3, step o: bf16 form of the 3 x 3 weight-gradient kernel in the mixed-precision training mode: step times (fp32 / bf16),
# the training tier's GPU tests in the default mode, and the full-size training test in the bf16 mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_o
mkdir -p $O
{
  for prec in f32 bf16; do
    echo "# FIERY_TRAIN_PRECISION=$prec"
    FIERY_TRAIN_PRECISION=$prec timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step
    FIERY_TRAIN_PRECISION=$prec timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --from-images 2>&1 | grep time_train_step
  done
} > $O/train_step.txt
cat $O/train_step.txt
FIERY_TRAIN_PRECISION=bf16 timeout 900 python -m pytest tests/test_train_graph.py -q -m gpu -x -k "reduce_the_loss" > $O/pytest_bf16_training.txt 2>&1; tail -2 $O/pytest_bf16_training.txt | cut -c1-200
timeout 900 python -m pytest tests/test_train_graph.py -q -m gpu -x > $O/pytest_training.txt 2>&1; tail -2 $O/pytest_training.txt | cut -c1-200
