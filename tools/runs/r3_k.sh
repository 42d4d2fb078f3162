#!/bin/bash
# Round 3, step k: the bf16 halo loop on 128-pixel tiles too (128 x 64, 128 x 128): microbenchmark with the tile height the
# operator measures (both are timed), forced 64 and forced 128; bench line; the bf16 GPU tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_k
mkdir -p $O
for tile in auto 64 128; do
  [ $tile = auto ] && unset FIERY_CONV_TILE_M || export FIERY_CONV_TILE_M=$tile
  echo "== tile height: $tile" >> $O/microbench_bf16.txt
  CONV_PRECISION=bf16 timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep "k3 s1" >> $O/microbench_bf16.txt
done
unset FIERY_CONV_TILE_M
cat $O/microbench_bf16.txt
timeout 300 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*' $O/bench_baseline_bf16.json
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16" > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
