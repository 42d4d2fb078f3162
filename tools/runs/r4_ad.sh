#!/bin/bash
# Round 4: round 3's pooling translation unit against today's (tuning hooks compiled out of the production kernel, sixteen
# completion counters), one box: kernel-trace durations, then alternating whole-op timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_ad
mkdir -p $O
cd /tmp
for v in r3pool today_clean today_nocounters r3pool_again today_clean_again; do
  lib=$R/fiery_amd/libfiery_hip.so; var="PY_CLEAN=1"
  case $v in r3pool*) lib=$R/tools/ab/libfiery_hip_r3pool.so; var="-";; today_nocounters) var="FIERY_POOL_NO_COUNTERS=1";; esac
  rm -rf /tmp/kt_$v
  FIERY_HIP_LIB=$lib ROUNDS=1 REPS=40 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $R/tools/runs/r4_pool_ab.py "$var" > $O/kt_$v.log 2>&1
  db=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$v.csv "round 4: pooling op, $v" 2>&1 | tail -1 > /dev/null
  echo "== $v"; grep -E "k_rank|k_voxel_pool|fillBuffer" $O/kernel_stats_$v.csv | cut -c1-160; grep "^\[0\]" $O/kt_$v.log | cut -c1-160
done 2>&1 | tee $O/pool_kernels.txt
cd $R
for rep in 1 2; do
  echo "== round 3 lift_splat.hip (rep $rep)"
  FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_r3pool.so ROUNDS=2 REPS=40 timeout 300 python tools/runs/r4_pool_ab.py "-" 2>&1 | grep "^\["
  echo "== today's (rep $rep)"
  ROUNDS=2 REPS=40 timeout 300 python tools/runs/r4_pool_ab.py "-" "PY_CLEAN=1" 2>&1 | grep "^\["
done | tee $O/pool_r3_vs_r4.txt
