#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool or lift or splat or voxel" > $O/pool4_pytest.txt 2>&1; tail -3 $O/pool4_pytest.txt
POOL_TILES=20480,40000 timeout 300 python tools/microbench.py pool --probe --reps 10 2>&1 | grep "pool\|probe" > $O/pool4.txt
cat $O/pool4.txt
