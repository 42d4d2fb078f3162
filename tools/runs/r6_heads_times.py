#!/usr/bin/env python
"""Round 6: the decoder-heads launch (64 -> 4 x 64 hidden -> 7 rows, 15 x 200 x 200) in its forms; with a -DW_EXP library: where its time is."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, HeadsOut, identity_chan_map       # noqa: E402

lib = native.get()
torch.manual_seed(0)
n, H, W, cin, hid = int(os.environ.get('N_IMG', 15)), 200, 200, 64, 64
n_outs, sig = [2, 1, 2, 2], [False, True, False, False]
x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
w1 = torch.randn(4 * hid, cin, 3, 3) / 24
op = ConvOp(lib, w1, identity_chan_map(cin), (cin // 8, 0), torch.ones(4 * hid), torch.zeros(4 * hid), 'cuda', act=native.ACT_RELU, tune=True)
groups = [i for i, o in enumerate(n_outs) for _ in range(o)]
op.attach_heads(torch.randn(7, hid) / 8, torch.zeros(7), groups, [sig[i] for i in groups])
results = [torch.zeros((n, o, H, W), device='cuda') for o in n_outs]
planes = [(res.data_ptr() + 4 * j * H * W, o * H * W) for res, o in zip(results, n_outs) for j in range(o)]
row = []
for form in (0, 'wino', 'wsplit', 0, 'wino', 'wsplit'):
    op.force_form = form
    for _ in range(3):
        op([x], HeadsOut(n, H, W, results[0]), head_planes=planes)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        op([x], HeadsOut(n, H, W, results[0]), head_planes=planes)
    e.record()
    e.synchronize()
    row.append(f'{form}: {s.elapsed_time(e) * 100:7.1f}')
print(f'heads 64 -> 256 -> 7 rows, {n} x {H} x {W}:  ' + '   '.join(row) + ' us')
