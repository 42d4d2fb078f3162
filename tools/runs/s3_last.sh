#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/last
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 60 python __graft_entry__.py --smoke 2>&1 | tail -1
