#!/usr/bin/env python
"""Round 6: the staged 3 x 3 weight-gradient kernel in its three modes (fp32 / bf16 operands / split) on training-step shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402

lib = native.get()
torch.manual_seed(0)
for cin, cout, n, H, W in ((64, 64, 4, 200, 200), (128, 128, 2, 200, 200), (64, 128, 2, 200, 200), (32, 32, 8, 200, 200), (128, 128, 10, 50, 50), (256, 256, 10, 25, 25)):
    x = torch.randn(n, H, W, cin, device='cuda')
    g = torch.randn(n, H, W, cout, device='cuda')
    row, ref = [], None
    for name, prec in (('f32', native.PRECISION_F32), ('bf16', native.PRECISION_BF16), ('split', native.PRECISION_F32_SPLIT)):
        for _ in range(2):
            dw = lib.conv_wgrad(x, g, cout, 3, 1, 1, prec)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            dw = lib.conv_wgrad(x, g, cout, 3, 1, 1, prec)
        e.record()
        e.synchronize()
        us = s.elapsed_time(e) * 100
        if ref is None:
            ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (cout, cin, 3, 3), g.permute(0, 3, 1, 2).double(), padding=1)
            ref = ref.reshape(cout, cin, 9).permute(0, 2, 1)
        err = ((dw[:, :, :cin].double() - ref).norm() / ref.norm()).item()
        flops = 2.0 * n * H * W * cin * cout * 9
        row.append(f'{name}: {us:7.1f} us ({flops / us / 1e6:5.1f} TFLOP/s, rel err {err:.1e})')
    print(f'{cin}->{cout} x{n} @{H}: ' + '   '.join(row), flush=True)
