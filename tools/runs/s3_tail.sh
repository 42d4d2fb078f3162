#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for p in 1 2 4 8; do echo "tail_parts=$p"; FIERY_POOL_TAIL_PARTS=$p timeout 300 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9 tile\|cold"; done
echo default; timeout 300 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9 tile\|cold"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline']['achieved'], b['roofline_pooling']['op_us_per_step'], b['roofline_pooling']['frac'])"; done
