#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2p5
mkdir -p $O
(
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pool or projection or splat" 2>&1 | tail -3
for rep in 1 2; do
echo "== baseline compact (default)"; timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
echo "== baseline dense plane"; FIERY_POOL_COMPACT=0 timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
done
echo "== pon compact"; POOL_PRESET=literature/pon_setting.yml timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
echo "== pon tiled"; FIERY_POOL_COMPACT=0 POOL_PRESET=literature/pon_setting.yml timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
echo "== fishing compact"; POOL_PRESET=literature/fishing_setting.yml timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
echo "== fishing tiled/dense"; FIERY_POOL_COMPACT=0 POOL_PRESET=literature/fishing_setting.yml timeout 200 python tools/microbench.py pool --reps 20 2>&1 | grep "^pool frames=9"
) 2>&1 | tee $O/pool5.txt
