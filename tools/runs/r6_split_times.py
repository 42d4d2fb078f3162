#!/usr/bin/env python
"""Round 6: launch times of the split tile form beside the fp32 forms on the step's shapes that have it (one box, alternating)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

lib = native.get()
torch.manual_seed(0)


def timed(run, reps=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


cases = [('stem 7x7/2 64->64 @200 x15', 15, 64, 64, 7, 2), ('64->64 3x3 @200 x3', 3, 64, 64, 3, 1), ('64->32 1x1 @200 x12', 12, 64, 32, 1, 1),
         ('32->32 3x3 @200 x4', 4, 32, 32, 3, 1), ('64->64 1x1 @200 x6', 6, 64, 64, 1, 1), ('64->96 1x1 @200 x6', 6, 64, 96, 1, 1)]
for name, n, cin, cout, k, stride in cases:
    x = Buf(torch.randn(n, 200, 200, cin, device='cuda'), n, 200, 200, cin)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', stride=stride, act=native.ACT_RELU, tune=True)
    ho, wo = op.out_hw(200, 200)
    out = Buf.alloc(n, ho, wo, cout, 'cuda')
    row = []
    for form in ([0] if cout % 64 else [64, 128, 'sk']) + ['split']:
        op.force_form = form
        us = timed(lambda: op([x], out))
        row.append(f'{form}: {us:7.1f}' + ('' if op.last_form == form else f' (ran {op.last_form})'))
    print(f'{name:28s} ' + '   '.join(row), flush=True)
# the Bottleneck tail with the next block's down-projection
n, mid, cout = 12, 32, 64
x = Buf(torch.randn(n, 200, 200, mid, device='cuda'), n, 200, 200, mid)
res = Buf(torch.randn(n, 200, 200, cout, device='cuda'), n, 200, 200, cout)
base = ConvOp(lib, torch.randn(mid, mid, 3, 3) / 17, identity_chan_map(mid), (4, 0), torch.ones(mid), torch.zeros(mid), 'cuda', act=native.ACT_RELU, tune=True)
base.chain_pointwise(torch.randn(cout, mid, 1, 1) / 6, torch.ones(cout), torch.zeros(cout), native.ACT_RELU)
op = base.chain_next(torch.randn(mid, cout, 1, 1) / 8, torch.ones(mid), torch.zeros(mid), native.ACT_RELU)
out, nxt = Buf.alloc(n, 200, 200, cout, 'cuda'), Buf.alloc(n, 200, 200, mid, 'cuda')
row = []
for form in (0, 'split', 0, 'split'):
    op.force_form = form
    us = timed(lambda: op([x], out, res=res, out3=nxt))
    row.append(f'{form}: {us:7.1f}' + ('' if op.last_form == form else f' (ran {op.last_form})'))
print(f'{"tail 32->32 + 64 + next @200 x12":28s} ' + '   '.join(row), flush=True)
