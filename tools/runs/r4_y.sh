#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_y
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "conv or gru or decoder or forward or hot_path or fixture" 2>&1 | tail -2
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== FIERY_CONV_DENSE_EPILOGUE=$v (rep $rep)"
    FIERY_CONV_DENSE_EPILOGUE=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-from-images --no-bf16-mode --no-secondary-configs --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['timed_mode']['frac'], d['roofline_pooling']['op_us_per_step'], d['roofline_pooling']['op_us_samples'])"
  done
done | tee $O/bench_ab2.txt
