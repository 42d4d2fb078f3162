#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/v2_pytest.txt 2>&1; tail -3 $O/v2_pytest.txt
for TM in 64 128; do
  FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_base.so FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --reps 10 > $O/v2_conv_base_tm$TM.txt 2>&1
  FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --reps 10 > $O/v2_conv_new_tm$TM.txt 2>&1
done
FIERY_BENCH_DUMP=$O/v2_launches.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/v2_bench.json 2> $O/v2_bench.err
paste -d'\n' $O/v2_conv_base_tm64.txt $O/v2_conv_new_tm64.txt | grep conv | cut -c1-110 | head -12
cut -c1-330 $O/v2_bench.json
