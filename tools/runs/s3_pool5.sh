#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or lift or splat or voxel or backward or autograd or seam" > $O/pytest_pool5.txt 2>&1; tail -3 $O/pytest_pool5.txt
(for t in 20480 40000; do
for cfg in "PIPE=1 B=7" "PIPE=0 B=7" "PIPE=1 B=14"; do
  eval $cfg
  echo "tile=$t pipe=$PIPE batch=$B"; POOL_TILES=$t FIERY_POOL_PIPE=$PIPE FIERY_POOL_BATCH=$B timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool frames=9 tile\|cold"
done; done
cd /tmp
for t in 20480 40000; do
  rm -rf /tmp/pmcx
  POOL_TILES=$t timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcx -o pmc -- python $GRAFT_REPO_ROOT/tools/microbench.py pool --reps 2 > /tmp/pmcx.log 2>&1
  echo "== tile=$t FETCH_SIZE"
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmcx/**/*.db" 2>&1 | grep "k_voxel_pool<"
done) 2>&1 | tee $O/pool_pipe.txt
