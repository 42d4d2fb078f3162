#!/usr/bin/env python
"""Round 6: exactly N hot-path steps and nothing else, for `rocprofv3 --kernel-trace --stats` - so that the stats' totals divided
by N ARE per-step kernel times (bench.py's own process also holds the tuner's trial launches, warm-up, instrumented steps, the
bf16 leg ...).  The convolution forms come from a table a previous process measured and saved:
  python tools/runs/r6_profile_steps.py tune  <forms.json>                      tunes (one eager pass per launch mode), saves
  rocprofv3 --kernel-trace --stats ... -- python tools/runs/r6_profile_steps.py run <forms.json> --steps 20 [--sample-streams] [--graph]
baseline.yml, 6 cameras x 3 frames, batch 3, fp32 - bench.py's headline workload and inputs."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import ops                                      # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.model import Fiery                              # noqa: E402
from fiery_amd.synthetic import make_inputs, make_lifted_features, randomise_weights   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('what', choices=['tune', 'run'])
ap.add_argument('forms')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--sample-streams', action='store_true')
ap.add_argument('--graph', action='store_true')
ap.add_argument('--precision', default='f32')
ap.add_argument('--config', default='baseline.yml')
ap.add_argument('--cams', type=int, default=0)
args = ap.parse_args()

dev = torch.device('cuda:0')
cfg = get_preset_cfg(args.config)
torch.manual_seed(0)
model = Fiery(cfg).eval()
randomise_weights(model)
model = model.to(dev)
model.conv_precision = args.precision
model.camera_matrix_mode = 'device'
B, rf, nf = 3, model.receptive_field, model.n_future
n_cam = args.cams or len(cfg.IMAGE.NAMES)
D = model.depth_channels
fh, fw = cfg.IMAGE.FINAL_DIM[0] // 8, cfg.IMAGE.FINAL_DIM[1] // 8
C = cfg.MODEL.ENCODER.OUT_CHANNELS
_, K, E, ego = make_inputs(B, rf + nf, n_cam, with_image=False, seed=0)
_, _, lifted = make_lifted_features(B * rf * n_cam, C, D, (fh, fw), seed=100)
lifted = lifted.view(B, rf, n_cam, C, D, fh, fw).to(dev)
K, E, ego = K.to(dev), E.to(dev), ego.to(dev)

with torch.no_grad():
    if args.what == 'tune':
        for streams in (False, True):
            model.sample_streams = streams
            model.bev_forward(lifted, K, E, ego)
            torch.cuda.synchronize()
        ops.save_form_table(args.forms)
        print(f'{len(ops.form_table())} measured forms -> {args.forms}')
    else:
        ops.load_form_table(ops.read_form_table(args.forms), frozen=True)
        model.sample_streams = args.sample_streams
        step = (lambda: model.bev_forward_graph(lifted, K, E, ego)) if args.graph else (lambda: model.bev_forward(lifted, K, E, ego))
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        print(f'{args.steps} steps, sample_streams={args.sample_streams}, graph={args.graph}')
