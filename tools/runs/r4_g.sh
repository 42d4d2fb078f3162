#!/bin/bash
# Round 4: the op without its memset dispatch (self-cleaning workspace) and with the four-lane prepass; GPU pooling tests; bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_g
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "voxel_pool or pooling or indices or lift or geometry" 2>&1 | tail -3
ROUNDS=3 REPS=40 timeout 600 python tools/runs/r4_pool_ab.py "FIERY_POOL_PREPASS_LANES=1" "-" "PY_CLEAN=1" 2>&1 | tee $O/pool_ab.txt
ROUNDS=2 REPS=40 timeout 300 python tools/runs/r4_pool_ab.py literature/pon_setting.yml "FIERY_POOL_PREPASS_LANES=1" "PY_CLEAN=1" 2>&1 | tee $O/pool_ab_pon.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-from-images --no-bf16-mode > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
