#!/bin/bash
# Round 5: Winograd per-kind kernels: GPU parity (forms test + hot path), kernel check, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv_forms or conv_igemm_real or hot_path_against_reference_fixture or hot_path_baseline_batch3 or per_sample_streams or graph_replay" 2>&1 | tail -5 | tee $O/pytest.txt
bash tools/runs/r5_e.sh
