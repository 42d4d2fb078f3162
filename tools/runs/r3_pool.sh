#!/bin/bash
# Round 3: the pooling op with the four-lane prepass + prepass-built occupancy words + 16-byte write-out (default) against the
# round-2 form, alternating on one box; then the GPU pooling tests; then the per-kernel trace of the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_pool
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "voxel_pool or pooling or indices" 2>&1 | tail -3
for rep in 1 2; do
  for cfg in "1 1" "0 1"; do
    set -- $cfg
    echo "== quad prepass=$1 prepass words=$2 (rep $rep)"
    FIERY_POOL_QUAD_PREPASS=$1 FIERY_POOL_PREPASS_WORDS=$2 timeout 300 python tools/microbench.py pool --reps 30 2>&1 | grep -E "^pool frames=9 (tile|cold)" 
  done
done | tee $O/pool_ab.txt
cd /tmp
rm -rf /tmp/kt_pool
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_pool -o kt -- python $R/tools/microbench.py pool --reps 30 > $O/kt.log 2>&1
db=$(find /tmp/kt_pool -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_pool_microbench.csv "round 3: rocprofv3 --kernel-trace --stats -- python tools/microbench.py pool --reps 30" 2>&1 | tail -2
grep -E "k_rank|k_voxel_pool|fillBuffer" $O/kernel_stats_pool_microbench.csv | cut -c1-200
