#!/bin/bash
# Round 4: the pooling op INSIDE the step (bench.py's roofline_pooling) - round 3's form (one-lane prepass, memset dispatch) against
# this round's, and the two ways of cutting the tail units, alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_z
mkdir -p $O
for rep in 1 2; do
  for v in "FIERY_POOL_PREPASS_LANES=1 FIERY_POOL_NO_CLEAN=1 FIERY_POOL_PART_RANGES=0" "FIERY_POOL_PART_RANGES=0" "FIERY_POOL_PART_RANGES=1"; do
    echo "== $v (rep $rep)"
    env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-from-images --no-bf16-mode --no-cpu-baseline --no-secondary-configs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline_pooling']['op_us_per_step'], d['roofline_pooling']['frac'])"
  done
done | tee $O/pool_in_step_ab.txt
