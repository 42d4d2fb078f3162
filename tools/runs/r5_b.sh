#!/bin/bash
# Round 5: register-chained Bottleneck tails (128 x 32 tile, lane = pixel): GPU parity subset + same-box A/B against round 4's library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "conv_igemm or hot_path_against_reference_fixture or bf16_conv_mode or hot_path_baseline_batch3" 2>&1 | tail -8 | tee $O/pytest.txt
TAG=r5_b LIBS="tools/ab/libfiery_hip_r4.so -" REPS=2 bash tools/runs/r5_ab.sh
TAG=r5_b_bf16 LIBS="tools/ab/libfiery_hip_r4.so -" REPS=1 EXTRA="--precision bf16" bash tools/runs/r5_ab.sh
