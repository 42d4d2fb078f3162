#!/usr/bin/env python
"""Round 5: stream-K on the GPU - the thing the simulator cannot see: partial tiles handed between workgroups on different
XCDs.  Real shapes, the stream-K form against the 128-pixel tile form of the same launch, repeated (stale partials of the
previous launch sit in every workspace slot; a partial read too early or from a stale L2 line is an O(1) error in a tile)."""
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
for cin, cout, n, H, W, k, reps in ((128, 128, 3, 200, 200, 3, 30), (128, 64, 3, 200, 200, 3, 30), (256, 256, 15, 25, 25, 3, 100),
                                    (64, 128, 3, 200, 200, 3, 30), (64, 64, 15, 100, 100, 3, 30), (128, 128, 15, 50, 50, 3, 60),
                                    (96, 64, 3, 200, 200, 1, 30)):
    x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=True)
    ref = Buf.alloc(n, H, W, cout, 'cuda')
    op.force_form = 128
    op([x], ref)
    op.force_form = 'sk'
    worst, times = 0.0, {}
    for form in (64, 128, 'sk'):
        op.force_form = form
        out = Buf.alloc(n, H, W, cout, 'cuda')
        op([x], out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            op([x], out)
        e.record()
        e.synchronize()
        times[form] = s.elapsed_time(e) * 100
    first = None
    for rep in range(reps):
        out = Buf.alloc(n, H, W, cout, 'cuda')
        out.tensor.fill_(float('nan'))
        op([x], out)
        err = (out.tensor - ref.tensor).abs().max().item()
        worst = max(worst, err if err == err else 1e9)
        if first is None:
            first = out.tensor.clone()
        elif not torch.equal(first, out.tensor):
            bad += 1
            print('   NOT bit-reproducible at rep', rep, (first - out.tensor).abs().max().item())
    flops = 2.0 * n * H * W * cin * k * k * cout
    ok = worst < 1e-4
    bad += not ok
    print(f'{cin:4d}->{cout:4d} k{k} {n}x{H}x{W}: stream-K vs tile form max |diff| {worst:.2e} over {reps} launches {"ok" if ok else "WRONG"};  '
          + '  '.join(f'{f}: {t:7.1f} us {flops / t / 1e6 / 157.3:.3f}' for f, t in times.items()))
print('FAILED' if bad else 'all stream-K launches equal the tile form')
sys.exit(1 if bad else 0)
