#!/usr/bin/env python
"""Round 5: launch time of the Winograd form on the step's shapes (one line; for same-box A/B of kernel variants)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

lib = native.get()
torch.manual_seed(0)
form = os.environ.get('FORM', 'wino')
cases = ((128, 128, 3, 200, 200), (128, 128, 3, 200, 200), (128, 64, 3, 200, 200), (64, 128, 3, 200, 200), (64, 64, 3, 200, 200),
         (64, 64, 15, 100, 100), (128, 128, 15, 50, 50), (256, 256, 15, 25, 25), (64, 256, 15, 200, 200))
row, tot = [], 0.0
for i, (cin, cout, n, H, W) in enumerate(cases):
    x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=True)
    op.force_form = form
    out = Buf.alloc(n, H, W, cout, 'cuda')
    for _ in range(3):
        op([x], out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        op([x], out)
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) * 50
    if i:
        row.append(f'{cin}->{cout}@{H}x{n}: {us:6.1f}')
        tot += us
print('  '.join(row) + f'   sum {tot:7.1f} us')
