#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/v3_pytest.txt 2>&1; tail -3 $O/v3_pytest.txt
for TM in 64 128; do
  FIERY_CONV_ALIGNED=0 FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --reps 10 > $O/v3_conv_generic_tm$TM.txt 2>&1
  FIERY_CONV_TILE_M=$TM timeout 300 python tools/microbench.py conv --reps 10 > $O/v3_conv_aligned_tm$TM.txt 2>&1
done
FIERY_BENCH_DUMP=$O/v3_launches.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/v3_bench.json 2> $O/v3_bench.err
for TM in 64 128; do echo "== TM $TM (generic | aligned)"; paste <(grep "^conv" $O/v3_conv_generic_tm$TM.txt | cut -c1-70) <(grep "^conv" $O/v3_conv_aligned_tm$TM.txt | cut -c36-70); done
cut -c1-330 $O/v3_bench.json
