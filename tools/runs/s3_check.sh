#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s3c
mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
FIERY_BENCH_DUMP=$O/launches.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/s3c/bench.json'))
print(b['value'], b['ms_per_step'], b['roofline']['achieved'], b['roofline_pooling'])
PY
FIERY_POOL_TILE=20480 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('tile 20480:', b['value'], b['roofline_pooling']['op_us_per_step'])"
timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "pool\|lift"
