#!/usr/bin/env python
"""Tuning aid: wall time of the full `Fiery.forward` (images in, trunk on stock PyTorch-ROCm) next to the hot path alone."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiery_amd.config import get_preset_cfg                      # noqa: E402
from fiery_amd.model import Fiery                                # noqa: E402
from fiery_amd.synthetic import make_inputs                      # noqa: E402
from tests.helpers import randomise_weights                      # noqa: E402


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device('cuda', 0)
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg).eval()
    randomise_weights(model)
    model = model.to(dev)
    B = 3
    image, K, E, ego = make_inputs(B, model.receptive_field + model.n_future, 6, with_image=True, seed=0)
    image, K, E, ego = image.to(dev), K.to(dev), E.to(dev), ego.to(dev)
    rf = model.receptive_field
    x = image[:, :rf].reshape(-1, 3, *cfg.IMAGE.FINAL_DIM).contiguous()
    with torch.no_grad():
        ms_full = timed(lambda: model(image, K, E, ego))
        try:
            ms_graph = timed(lambda: model.forward_graph(image, K, E, ego))
        except Exception as e:                      # noqa: BLE001
            ms_graph = float('nan')
            print('graph capture failed:', repr(e)[:200])
        ms_trunk = timed(lambda: model.encoder.trunk_endpoints(x))
        ms_trunk_hip = timed(lambda: model.engine().trunk_endpoints(x))
        model.hip_trunk = False
        ms_full_torch_trunk = timed(lambda: model(image, K, E, ego))
        model.hip_trunk = True
        deep, shallow = model.encoder.trunk_endpoints(x)
        ms_head = timed(lambda: model.engine().lift_head(deep, shallow))
        ms_head_torch = timed(lambda: model.encoder.depth_layer(model.encoder.get_features(x))) - ms_trunk
    print(f'batch {B}: forward from images {ms_full:.2f} ms with the trunk on the engine ({ms_graph:.2f} ms replayed from a hipGraph), {ms_full_torch_trunk:.2f} ms with it on PyTorch-ROCm  '
          f'(trunk alone: engine {ms_trunk_hip:.2f} ms, PyTorch-ROCm {ms_trunk:.2f} ms for {x.shape[0]} images; '
          f'lift head on the engine {ms_head:.2f} ms, on PyTorch-ROCm {ms_head_torch:.2f} ms)')


if __name__ == '__main__':
    main()
