#!/bin/bash
# Round 5, last state: the whole GPU suite, smoke(), and the default bench line at HEAD
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_head
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5_head/bench.json'))
print(d['value'], d['unit'], 'ms', d['ms_per_step'], 'conv frac', d['roofline']['frac'], 'pool frac', d['roofline_pooling']['frac'],
      'from images', d['forward_from_images']['ms_per_step'], 'parity', d['parity_literal_1e-4'], 'bf16', (d.get('bf16_mode') or {}).get('value'))
PY
