#!/bin/bash
# Session 3: packed-fp32 pooling row loop + pooling backward - parity first, then timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or lift or splat or voxel or backward or autograd or seam" > $O/pytest_pool.txt 2>&1; tail -5 $O/pytest_pool.txt
timeout 300 python tools/microbench.py pool --probe --reps 10 > $O/pool.txt 2>&1
grep "pool\|probe\|lift" $O/pool.txt
for b in 16 8; do
  echo "FIERY_POOL_BATCH=$b"; FIERY_POOL_BATCH=$b timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "^pool frames=9 tile"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/s3b/bench.json'))
print(b['value'], b['roofline']['achieved'], b['roofline_pooling'])
PY
