#!/bin/bash
# Round 3, step n: bf16-operand convolutions in the training graph (FIERY_TRAIN_PRECISION=bf16: forward + input gradient on the
# bf16 kernels, weight gradient fp32): step time from the lifted features and from images, and the full-size training test
# (loss decreases over SGD steps, gradients finite) in that mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_n
mkdir -p $O
{
  for prec in f32 bf16; do
    echo "# FIERY_TRAIN_PRECISION=$prec"
    FIERY_TRAIN_PRECISION=$prec timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step
    FIERY_TRAIN_PRECISION=$prec timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --from-images 2>&1 | grep time_train_step
  done
} > $O/train_step.txt
cat $O/train_step.txt
FIERY_TRAIN_PRECISION=bf16 timeout 900 python -m pytest tests/test_train_graph.py -q -m gpu -x -k "reduce_the_loss or bf16_operand" > $O/pytest_bf16_training.txt 2>&1; tail -3 $O/pytest_bf16_training.txt | cut -c1-200
