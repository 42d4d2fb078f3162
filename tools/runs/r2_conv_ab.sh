#!/bin/bash
# conv microbench (fp32) + the bench line, for A/B of a tile-shape change
mkdir -p gpurun_out/r2_conv
timeout 600 python tools/microbench.py conv --reps 20 > gpurun_out/r2_conv/microbench.txt 2>&1
grep "^conv" gpurun_out/r2_conv/microbench.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-from-images > gpurun_out/r2_conv/bench.json 2> gpurun_out/r2_conv/bench.err
cut -c1-260 gpurun_out/r2_conv/bench.json
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "forward or whole or graph or conv" -x > gpurun_out/r2_conv/pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/r2_conv/pytest.txt | tail -1
