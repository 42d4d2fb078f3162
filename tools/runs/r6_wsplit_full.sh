#!/bin/bash
# default bench line (parity over three draws) with the split form among the candidates, then without; GPU parity subset
mkdir -p gpurun_out/wsplit; O=gpurun_out/wsplit
timeout 1200 python bench.py --steps 20 --warmup 3 --no-secondary-configs --no-from-images --no-bf16-mode > $O/full_split1.json 2>$O/full_err1.txt
FIERY_CONV_WINOGRAD_SPLIT=0 timeout 1200 python bench.py --steps 20 --warmup 3 --no-secondary-configs --no-from-images --no-bf16-mode --no-cpu-baseline > $O/full_split0.json 2>$O/full_err0.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hot_path or conv or graph or streams" > $O/pytest_subset.txt 2>&1; tail -4 $O/pytest_subset.txt
