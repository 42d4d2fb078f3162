#!/bin/bash
# Round 3, step j: the halo loop in fp32 (FIERY_CONV_HALO_F32=1) against the scalar-addressed loop: microbenchmark, bench line,
# the hot-path parity tests with the switch on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-r3_j}
mkdir -p $O
for halo in 1 0; do
  export FIERY_CONV_HALO_F32=$halo
  echo "== FIERY_CONV_HALO_F32=$halo" >> $O/microbench_f32.txt
  timeout 300 python tools/microbench.py conv --reps 20 >> $O/microbench_f32.txt 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --no-from-images --no-cpu-baseline > $O/bench_fp32_halo$halo.json 2>> $O/bench.err
  grep -h -o '"value": [0-9.]*' $O/bench_fp32_halo$halo.json
  grep -h -o '"roofline": {[^}]*}' $O/bench_fp32_halo$halo.json | cut -c1-200
done
grep -v amdgpu.ids $O/microbench_f32.txt
export FIERY_CONV_HALO_F32=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch3 or reference_fixture or other_reference_configs or conv_igemm_real_shapes or graph_replay or per_sample" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
