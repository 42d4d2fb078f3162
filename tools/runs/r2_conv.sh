#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
for TM in 64 128; do
  FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --clk --reps 10 > $O/conv_phase_tm$TM.txt 2>&1
  FIERY_CONV_PRIO=1 FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --reps 10 > $O/conv_prio_tm$TM.txt 2>&1
done
head -8 $O/conv_phase_tm64.txt; head -4 $O/conv_prio_tm64.txt
