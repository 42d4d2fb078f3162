#!/bin/bash
# Round 3, step l: the chained 1x1 GEMMs of the Bottleneck tails on bf16 MFMAs in the bf16 mode: bench lines (baseline, pon,
# lyft-7) and the bf16 GPU tests (errors against the fp32 oracle, argmax agreement).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_l
mkdir -p $O
FIERY_BENCH_DUMP=$O/launches_bf16.json timeout 300 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --config literature/pon_setting.yml --no-from-images --no-cpu-baseline > $O/bench_pon_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --config lyft/baseline.yml --cams 7 --no-from-images --no-cpu-baseline > $O/bench_lyft7_bf16.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*' $O/bench_baseline_bf16.json $O/bench_pon_bf16.json $O/bench_lyft7_bf16.json | grep -v "0\.[0-9]*$"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16" > $O/pytest_subset.txt 2>&1; tail -12 $O/pytest_subset.txt | cut -c1-200
