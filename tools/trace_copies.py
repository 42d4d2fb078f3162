#!/usr/bin/env python
"""Tuning aid: which host-side calls of one eager step put device-to-device copies / fills / ATen kernels on the GPU
(everything that is not a fiery:: kernel).  usage: python tools/trace_copies.py [--no-sample-streams]"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fiery_amd.config import get_preset_cfg                      # noqa: E402
from fiery_amd.model import Fiery                                # noqa: E402
from fiery_amd.synthetic import make_inputs, make_lifted_features  # noqa: E402
from tests.helpers import randomise_weights                      # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg).eval()
    randomise_weights(model)
    model = model.to(dev)
    model.sample_streams = '--no-sample-streams' not in sys.argv
    B, rf, nf, n_cam, D = 3, model.receptive_field, model.n_future, 6, model.depth_channels
    fh, fw = 28, 60
    _, K, E, ego = make_inputs(B, rf + nf, n_cam, with_image=False, seed=0)
    _, _, lifted = make_lifted_features(B * rf * n_cam, 64, D, (fh, fw), seed=100)
    lifted = lifted.view(B, rf, n_cam, 64, D, fh, fw).to(dev)
    K, E, ego = K.to(dev), E.to(dev), ego.to(dev)
    with torch.no_grad():
        for _ in range(3):
            model.bev_forward(lifted, K, E, ego)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            model.bev_forward(lifted, K, E, ego)
            torch.cuda.synchronize()
    names = Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA and 'fiery::' not in ev.name:
            names[ev.name[:90]] += 1
    print('non-fiery GPU activities of one eager step:')
    for k, v in names.most_common(20):
        print(f'  {v:4d}  {k}')
    sites = Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in ('aten::copy_', 'aten::zeros', 'aten::fill_', 'aten::zero_',
                                                                             'aten::cat', 'aten::clone', 'aten::contiguous'):
            frames = [f for f in (ev.stack or []) if 'fiery_amd' in f]
            sites[(ev.name, frames[0] if frames else '?')] += 1
    print('host call sites:')
    for (name, site), v in sites.most_common(40):
        print(f'  {v:4d}  {name:18s} {site}')


if __name__ == '__main__':
    main()
