#!/usr/bin/env python
"""Round 5: the Winograd F(2x2, 3x3) form on the GPU - real shapes against the direct 128-pixel tile form (max |diff| relative to
the output scale), and the launch time of every form (64 / 128 pixel tiles, stream-K, Winograd)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

lib = native.get()
torch.manual_seed(0)
bad = 0
cases = ((128, 128, 3, 200, 200), (128, 128, 3, 200, 200), (128, 64, 3, 200, 200), (64, 128, 3, 200, 200), (64, 64, 3, 200, 200),
         (64, 64, 15, 100, 100), (128, 128, 15, 50, 50), (256, 256, 15, 25, 25), (64, 256, 15, 200, 200), (32, 64, 12, 200, 200))
for cin, cout, n, H, W in cases:
    x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=True)
    ref = Buf.alloc(n, H, W, cout, 'cuda')
    op.force_form = 128
    op([x], ref)
    times, err = {}, None
    for form in (64, 128, 'sk', 'wino'):
        op.force_form = form
        out = Buf.alloc(n, H, W, cout, 'cuda')
        out.tensor.fill_(float('nan'))
        op([x], out)
        torch.cuda.synchronize()
        if form == 'wino':
            err = (out.tensor - ref.tensor).abs().max().item()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            op([x], out)
        e.record()
        e.synchronize()
        times[form] = s.elapsed_time(e) * 100
    flops = 2.0 * n * H * W * cin * 9 * cout
    ok = err == err and err < 2e-5 * max(1.0, ref.tensor.abs().max().item())
    bad += not ok
    print(f'{cin:4d}->{cout:4d} {n:2d}x{H}x{W}: Winograd vs direct max |diff| {err:.2e} (scale {ref.tensor.abs().max().item():.2f}) {"ok" if ok else "WRONG"};  '
          + '  '.join(f'{f}: {t:7.1f} us ({flops / t / 1e6:6.1f} alg. TFLOP/s)' for f, t in times.items()), flush=True)
print('FAILED' if bad else 'all Winograd launches within 2e-5 of the direct form')
sys.exit(1 if bad else 0)
