#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2c1
mkdir -p $O
(timeout 300 python tools/runs/r2_probe_planes.py 10 2>&1 | grep planes2) | tee $O/probe_planes.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -80 | tee $O/gputests.txt
timeout 200 python tools/trace_copies.py 2>&1 | tail -60 | tee $O/trace_copies.txt
