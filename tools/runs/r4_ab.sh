#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_ab
mkdir -p $O
cd /tmp
for v in r3pool today; do
  lib=$R/fiery_amd/libfiery_hip.so; [ $v = r3pool ] && lib=$R/tools/ab/libfiery_hip_r3pool.so
  var="PY_CLEAN=1"; [ $v = r3pool ] && var="-"
  rm -rf /tmp/kt_$v
  FIERY_HIP_LIB=$lib ROUNDS=1 REPS=40 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $R/tools/runs/r4_pool_ab.py "$var" > $O/kt_$v.log 2>&1
  db=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$v.csv "round 4: pooling op, $v" 2>&1 | tail -1
  echo "== $v"; grep -E "k_rank|k_voxel_pool|fillBuffer" $O/kernel_stats_$v.csv | cut -c1-160
done
