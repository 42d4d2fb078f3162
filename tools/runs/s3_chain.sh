#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hot_path or reference_configs or graph or streams or conv_igemm" 2>&1 | tail -3
for c in 1 0; do echo "FIERY_CHAIN_NEXT=$c"; FIERY_CHAIN_NEXT=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-images 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b['roofline']['achieved'], b['roofline']['launches'], b['roofline']['kernel_ms_per_step'])"; done
