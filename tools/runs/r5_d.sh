#!/bin/bash
# Round 5: the pooling op's ceiling on this box (tools/probe/pool_ceiling.hip), next to the op's own bracket from bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_d
mkdir -p $O
timeout 300 tools/probe/_bin/pool_ceiling 9 0.9409 2>&1 | tee $O/pool_ceiling.txt | tail -45
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs 2>/dev/null > $O/b.json
python - <<PY | tee -a $O/pool_ceiling.txt
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); rp=d['roofline_pooling']
print('same box, bench.py: pooling op %.1f us (%s), %.1f MB algorithmic, kept fraction %.4f, frac of 8 TB/s %.4f' % (rp['op_us_per_step'], rp['op_us_samples'], rp['algorithmic_mb_per_step'], rp['kept_fraction'], rp['frac']))
PY
