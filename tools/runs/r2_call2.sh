#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2c2
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -90 | tee $O/gputests.txt
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee $O/bench.txt
FIERY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --layout frames --steps 10 --warmup 2 --no-cpu-baseline --no-from-images 2>&1 | tail -3 | tee $O/bench_frames.txt
timeout 200 python tools/trace_copies.py 2>&1 | tail -30 | tee $O/trace_copies.txt
