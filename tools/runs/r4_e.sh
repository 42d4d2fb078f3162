#!/bin/bash
# Round 4: four-lane prepass against the one-lane prepass; tail variants; kernel durations from a trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_e
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "voxel_pool or pooling or indices" 2>&1 | tail -3
ROUNDS=3 REPS=40 timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_PREPASS_LANES=1,FIERY_POOL_PERSISTENT=0" "FIERY_POOL_PERSISTENT=0" "-" "FIERY_POOL_NO_DRAW=1,FIERY_POOL_TAIL_PARTS=4" "FIERY_POOL_TAIL_PARTS=8" 2>&1 | tee $O/pool_ab.txt
cd /tmp
rm -rf /tmp/kt_pool
ROUNDS=1 REPS=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_pool -o kt -- python $R/tools/runs/r4_pool_ab.py "FIERY_POOL_PREPASS_LANES=1,FIERY_POOL_PERSISTENT=0" "-" > $O/kt.log 2>&1
db=$(find /tmp/kt_pool -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_pool_ab.csv "round 4: rocprofv3 --kernel-trace --stats -- python tools/runs/r4_pool_ab.py (one-lane prepass + per-item grid, then defaults)" 2>&1 | tail -2
grep -E "k_rank|k_voxel_pool|fillBuffer" $O/kernel_stats_pool_ab.csv | cut -c1-200
