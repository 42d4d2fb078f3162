#!/bin/bash
# Round 6: GPU pooling / geometry / whole-path parity subset + the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_f}
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-pool or geometry or hot_path or lift_splat or projection}" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python bench.py ${BENCH_ARGS:---no-bf16-mode --no-secondary-configs --no-from-images} > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
o = os.environ.get('TAG', 'r6_f')
d = json.loads(open(f'gpurun_out/{o}/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline_pooling']['frac'], d['roofline_pooling']['op_us_samples'], d['roofline'].get('frac'))
PY
