#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or lift or splat or voxel or backward or autograd or seam" 2>&1 | tail -3
(for cfg in "PLANE=1 B=7" "PLANE=1 B=8" "PLANE=1 B=16" "PLANE=0 B=7"; do
  eval $cfg
  echo "plane=$PLANE batch=$B"; FIERY_POOL_PLANE=$PLANE FIERY_POOL_BATCH=$B timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool frames=9 tile\|cold"
done
cd /tmp; rm -rf /tmp/pmcx
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcx -o pmc -- python $GRAFT_REPO_ROOT/tools/microbench.py pool --reps 2 > /tmp/pmcx.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmcx/**/*.db" 2>&1 | grep "k_voxel_pool\|k_rank") 2>&1 | tee $O/pool_plane.txt
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline_pooling']['op_us_per_step'], b['roofline_pooling']['frac'])"
