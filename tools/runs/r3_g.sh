#!/bin/bash
# Round 3, step g: the engine branches no YAML uses (GPU test), the bf16 form with its A tile rounded at LDS-store time -
# microbenchmark and bench line, 4 against 3 wavefronts per SIMD (tools/ab/libfiery_hip_bf16w3.so = -DFIERY_BF16_WAVES=3).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "temporal_model_branches or bf16 or batch3" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
for lib in default w3; do
  [ $lib = w3 ] && export FIERY_HIP_LIB=$R/tools/ab/libfiery_hip_bf16w3.so || unset FIERY_HIP_LIB
  echo "== $lib" >> $O/microbench_bf16.txt
  CONV_PRECISION=bf16 timeout 300 python tools/microbench.py conv --reps 20 >> $O/microbench_bf16.txt 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16_$lib.json 2>> $O/bench.err
  grep -h -o '"value": [0-9.]*' $O/bench_baseline_bf16_$lib.json
done
unset FIERY_HIP_LIB
cat $O/microbench_bf16.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-from-images --no-cpu-baseline > $O/bench_fp32.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*' $O/bench_fp32.json
