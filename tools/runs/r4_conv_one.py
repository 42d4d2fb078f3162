#!/usr/bin/env python
"""Round 4: ONE conv shape, a few launches - the thing to put under rocprofv3 --pmc.
  python tools/runs/r4_conv_one.py [cin cout n H W [k]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

a = [int(v) for v in sys.argv[1:]]
cin, cout, n, H, W = (a + [128, 128, 3, 192, 256])[:5] if len(a) >= 5 else (128, 128, 3, 192, 256)
k = a[5] if len(a) > 5 else 3
lib = native.get()
x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=False)
out = Buf.alloc(n, H, W, cout, 'cuda')
for _ in range(6):
    op([x], out)
torch.cuda.synchronize()
print('done', cin, cout, n, H, W, k)
