#!/bin/bash
# Round 5: Winograd, tile blocks per workgroup (the next block's prologue inside this block's last stage): 1 / 2 / 4 / 8, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_s
mkdir -p $O
timeout 60 python tools/runs/r5_wino_times.py > /dev/null 2>&1
for rep in 1 2; do for b in 1 2 4 8; do
  echo "== $b blocks per workgroup (rep $rep)"
  FIERY_WINOGRAD_BPW=$b timeout 120 python tools/runs/r5_wino_times.py 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $O/times.txt
FIERY_WINOGRAD_BPW=4 timeout 200 python tools/runs/r5_wino_check.py 2>&1 | tail -3 | tee $O/check.txt
