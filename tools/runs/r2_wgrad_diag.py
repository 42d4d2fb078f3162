"""Diagnostic: fiery_conv_wgrad (staged 3x3 form) against torch autograd on the GPU for a ladder of shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from fiery_amd import native
lib = native.get()
for n, cin, cout, h, w in [(1, 64, 64, 1, 8), (1, 64, 64, 4, 50), (1, 32, 32, 4, 50), (2, 64, 64, 9, 14), (1, 64, 64, 4, 200),
                           (1, 64, 64, 40, 50), (2, 64, 64, 200, 200), (2, 128, 64, 200, 200), (8, 32, 32, 200, 200), (10, 256, 256, 25, 25)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g).cuda().requires_grad_()
    y = F.conv2d(x.double(), wt.double(), padding=1)
    gy = torch.randn(y.shape, generator=g).cuda()
    (want,) = torch.autograd.grad(y, wt, gy.double())
    xb = x.permute(0, 2, 3, 1).contiguous(); gb = gy.permute(0, 2, 3, 1).contiguous()
    dw = lib.conv_wgrad(xb, gb, cout, 3, 1, 1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        dw = lib.conv_wgrad(xb, gb, cout, 3, 1, 1)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 200
    got = dw.permute(0, 2, 1).reshape(cout, cin, 3, 3).double()
    err = (got - want).abs()
    flops = 2.0 * n * h * w * cin * cout * 9
    print((n, cin, cout, h, w), 'max err %.3e (scale %.1f)' % (err.max().item(), want.abs().max().item()), '%.1f us (incl. memset) %.1f TFLOP/s' % (us, flops / us / 1e6), flush=True)
