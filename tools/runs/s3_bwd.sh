#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or bwd or autograd or seam or adjoint" 2>&1 | tail -3
timeout 300 python tools/microbench.py pool --reps 10 2>&1 | grep "bwd"
