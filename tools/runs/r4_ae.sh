#!/bin/bash
# Round 4: calibration table on the GPU - its tests, the geometry / pooling parity tests in the new default mode, a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_ae
mkdir -p $O
timeout 600 python -m pytest tests/test_calibration_table.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench.json 2> $O/bench.err
cut -c1-330 $O/bench.json
FIERY_CAMERA_MATRICES=device timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_device_mode.json 2>> $O/bench.err
cut -c1-330 $O/bench_device_mode.json
tail -5 $O/bench.err
