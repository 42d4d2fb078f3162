#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3b
mkdir -p $O
cd /tmp
for t in 20480 40000; do
for b in 7 16; do
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
    name=$(echo $ctr | tr ' ' '_')
    rm -rf /tmp/pmcx
    POOL_TILES=$t FIERY_POOL_BATCH=$b timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmcx -o pmc -- python $GRAFT_REPO_ROOT/tools/microbench.py pool --reps 2 > /tmp/pmcx.log 2>&1
    echo "== tile=$t batch=$b $ctr"
    python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmcx/**/*.db" 2>&1 | grep "k_voxel_pool"
  done
done; done 2>&1 | tee $O/pool_pmc.txt
