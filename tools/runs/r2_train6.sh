#!/bin/bash
mkdir -p gpurun_out/r2_train
export TMPDIR=/tmp
for mode in torch32 hipB2 hipnocudnn; do
  echo "== $mode" >> gpurun_out/r2_train/diag2.txt
  timeout 300 python tools/runs/r2_train_diag2.py $mode >> gpurun_out/r2_train/diag2.txt 2>&1
  echo "rc=$?" >> gpurun_out/r2_train/diag2.txt
done
grep -v "^  File\|^Extension\|^$\|amdgpu.ids" gpurun_out/r2_train/diag2.txt | tail -50
