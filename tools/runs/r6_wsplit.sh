#!/bin/bash
# the split Winograd form: per-op parity (GPU tests) and launch times beside the fp32 form, alternating
mkdir -p gpurun_out/wsplit; O=gpurun_out/wsplit
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wsplit" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for r in 1 2; do for f in wino wsplit; do echo -n "$f  "; FORM=$f timeout 200 python tools/runs/r5_wino_times.py 2>&1 | tail -1; done; done | tee $O/times.txt
