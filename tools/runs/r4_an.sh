#!/bin/bash
# Round 4: pooling op bracket (stable to +-1 us since the bracket fix) for the tail's part count and the prepass rows per wavefront
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_an
mkdir -p $O
for v in "-" "FIERY_POOL_TAIL_PARTS=2" "FIERY_POOL_TAIL_PARTS=3" "FIERY_POOL_TAIL_PARTS=6" "FIERY_POOL_TAIL_PARTS=8" "FIERY_POOL_NT=0" "-"; do
  var=""; [ "$v" != "-" ] && var="$v"
  env $var timeout 60 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs 2>/dev/null > $O/b.json
  python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
rp=d['roofline_pooling']
print('%-28s pooling op %.1f us (%.4f)  samples %s' % ('$v', rp['op_us_per_step'], rp['frac'], rp['op_us_samples']))
PY
done 2>&1 | tee $O/summary.txt
