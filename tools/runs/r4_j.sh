#!/bin/bash
# Round 4: persistent conv workgroups (a workgroup takes several pixel tiles) against a workgroup per tile, one box, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "conv" 2>&1 | tail -2
for rep in 1 2; do
  for v in 0 1; do
    echo "== FIERY_CONV_PERSISTENT=$v (rep $rep)"
    FIERY_CONV_PERSISTENT=$v timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep "^conv"
  done
done | tee $O/conv_persistent_ab.txt
for v in 0 1 0 1; do
  echo "== FIERY_CONV_PERSISTENT=$v"
  FIERY_CONV_PERSISTENT=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-from-images --no-bf16-mode --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['timed_mode'])"
done | tee $O/bench_ab.txt
