#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3b
mkdir -p $O
(for lib in "" $GRAFT_REPO_ROOT/tools/ab/libfiery_hip_nont.so; do
for t in 20480 40000; do
for b in 4 7 14; do
  echo "lib=$lib tile=$t FIERY_POOL_BATCH=$b"; FIERY_HIP_LIB=$lib POOL_TILES=$t FIERY_POOL_BATCH=$b timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool frames=9 tile"
done; done; done
cd /tmp
for lib in "" $GRAFT_REPO_ROOT/tools/ab/libfiery_hip_nont.so; do
  rm -rf /tmp/pmcx
  FIERY_HIP_LIB=$lib POOL_TILES=40000 FIERY_POOL_BATCH=7 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcx -o pmc -- python $GRAFT_REPO_ROOT/tools/microbench.py pool --reps 2 > /tmp/pmcx.log 2>&1
  echo "== lib=$lib tile=40000 batch=7 FETCH_SIZE"
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmcx/**/*.db" 2>&1 | grep "k_voxel_pool<"
done) 2>&1 | tee $O/pool_nt_ab.txt
