#!/bin/bash
# Round 4: the fp32 matrix pipe's ceiling on this box (pure MFMA, random operands, every SIMD), then the conv microbench on the same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_h
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w tools/probe/mfma_ceiling.hip -o /tmp/mfma_ceiling && timeout 120 /tmp/mfma_ceiling 2>&1 | tee $O/mfma_ceiling.txt
timeout 300 python tools/microbench.py conv --reps 20 2>&1 | tee $O/conv_microbench.txt
timeout 120 /tmp/mfma_ceiling 2>&1 | tee $O/mfma_ceiling_after.txt
