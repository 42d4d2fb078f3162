#!/bin/bash
mkdir -p gpurun_out/r2_images
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "image_preparation or label" > gpurun_out/r2_images/pytest.txt 2>&1
grep -E "^E  |passed|failed" gpurun_out/r2_images/pytest.txt | cut -c1-200 | tail -5
python - <<'PY' > gpurun_out/r2_images/timing.txt 2>&1
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from fiery_amd.images import resize_crop_normalise
from oracle.images import prepare
g = torch.Generator().manual_seed(0)
images = torch.randint(0, 256, (42, 900, 1600, 3), generator=g, dtype=torch.uint8)        # one sample: 6 cameras x 7 frames
dims, crop = (480, 270), (0, 46, 480, 270)
t0 = time.time(); want = prepare(images.numpy()[:6], dims, crop); pil_ms = (time.time() - t0) / 6 * 1e3
for where in ('cpu', 'cuda'):
    x = images.to(where)
    for _ in range(2):
        out = resize_crop_normalise(x, dims, crop)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        out = resize_crop_normalise(x, dims, crop)
    torch.cuda.synchronize()
    print('42 frames 900x1600 -> 224x480, input on %s: %.2f ms per sample (%.3f ms per image)' % (where, (time.time() - t0) / 10 * 1e3, (time.time() - t0) / 420 * 1e3))
print('Pillow resize + crop + ToTensor + Normalize on one host core: %.2f ms per image (%.0f ms per sample)' % (pil_ms, pil_ms * 42))
print('equal to Pillow:', torch.equal(out[:6].cpu(), want))
PY
grep -v amdgpu gpurun_out/r2_images/timing.txt
