#!/bin/bash
mkdir -p gpurun_out/split; O=gpurun_out/split
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "split_tile" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 600 python tools/runs/r6_split_times.py 2>&1 | tee $O/times.txt
