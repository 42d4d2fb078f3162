#!/usr/bin/env python
"""Round 6: does the gap between two hipGraph replays matter?  One step per graph replayed 2N times against two steps per graph
replayed N times (baseline.yml, batch 3, one stream per sample), wall time per step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import ops                                      # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.model import Fiery                              # noqa: E402
from fiery_amd.synthetic import make_inputs, make_lifted_features, randomise_weights   # noqa: E402

dev = torch.device('cuda:0')
cfg = get_preset_cfg('baseline.yml')
torch.manual_seed(0)
model = Fiery(cfg).eval()
randomise_weights(model)
model = model.to(dev)
model.camera_matrix_mode = 'device'
model.sample_streams = True
B, rf, nf, n_cam, D = 3, model.receptive_field, model.n_future, 6, model.depth_channels
_, K, E, ego = make_inputs(B, rf + nf, n_cam, with_image=False, seed=0)
_, _, lifted = make_lifted_features(B * rf * n_cam, 64, D, (28, 60), seed=100)
lifted = lifted.view(B, rf, n_cam, 64, D, 28, 60).to(dev)
K, E, ego = K.to(dev), E.to(dev), ego.to(dev)
with torch.no_grad():
    for _ in range(2):
        model.bev_forward(lifted, K, E, ego)
    torch.cuda.synchronize()
    graphs = {}
    for per in (1, 2, 4):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ops.prepare_capture(dev)):
            for _ in range(per):
                out = model.bev_forward(lifted, K, E, ego)
        graphs[per] = g
    for rep in range(3):
        for per, g in graphs.items():
            n = 40 // per
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (n * per) * 1e3
            print(f'[{rep}] {per} step(s) per graph: {ms:.3f} ms per step = {3 / ms * 1e3:.1f} samples/s', flush=True)
