#!/bin/bash
mkdir -p gpurun_out/r2_labels
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "label" > gpurun_out/r2_labels/pytest.txt 2>&1
grep -E "^E  |passed|failed" gpurun_out/r2_labels/pytest.txt | cut -c1-200 | tail -5
python - <<'PY' > gpurun_out/r2_labels/timing.txt 2>&1
import sys, time, torch
sys.path.insert(0, '.')
from fiery_amd.labels import convert_instance_mask_to_center_and_offset_label
from tests.test_kernels_sim_aux import _label_blobs
ids, ego = _label_blobs(11, 7, 200, 200, 30)
for where in ('cpu', 'cuda'):
    a, b = ids.to(where), ego.to(where)
    for _ in range(3):
        out = convert_instance_mask_to_center_and_offset_label(a, b, 30, spatial_extent=(50.0, 50.0))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        out = convert_instance_mask_to_center_and_offset_label(a, b, 30, spatial_extent=(50.0, 50.0))
    torch.cuda.synchronize()
    print('instance labels, 7 x 200 x 200, 30 instances, tensors on %s: %.2f ms per sample' % (where, (time.time() - t0) / 20 * 1e3))
PY
grep "instance labels" gpurun_out/r2_labels/timing.txt
