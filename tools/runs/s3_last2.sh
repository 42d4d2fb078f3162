#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/last
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/last/pytest_gpu.txt 2>&1; tail -2 gpurun_out/last/pytest_gpu.txt
