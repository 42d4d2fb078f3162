#!/bin/bash
# Round 3, step h: the bf16 form's halo loop (3 x 3 layers on 64-pixel tiles: the A tile fetched once per channel group)
# against the scalar-addressed loop (FIERY_CONV_HALO=0): GPU tests of the bf16 form, microbenchmark, bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-r3_h}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "temporal_model_branches or bf16" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
for halo in 1 0; do
  export FIERY_CONV_HALO=$halo
  echo "== FIERY_CONV_HALO=$halo" >> $O/microbench_bf16.txt
  CONV_PRECISION=bf16 timeout 300 python tools/microbench.py conv --reps 20 >> $O/microbench_bf16.txt 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16_halo$halo.json 2>> $O/bench.err
  grep -h -o '"value": [0-9.]*' $O/bench_baseline_bf16_halo$halo.json
done
unset FIERY_CONV_HALO
grep -v amdgpu.ids $O/microbench_bf16.txt
