#!/usr/bin/env python
"""Round 4: timeline of one conv launch's workgroups (tuning build, FIERY_HIP_LIB=tools/ab/libfiery_hip_tuning.so): the 100 MHz
wall clock at every tile's entry / K loop start / K loop end / end, and where it ran (XCD, CU, SIMD)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

DEV = 'cuda:0'
lib = native.get()
out_dir = sys.argv[1] if len(sys.argv) > 1 else None
CASES = [(3, 1, 128, 128, 3, 192, 256), (3, 1, 128, 128, 3, 200, 200), (3, 1, 128, 64, 3, 200, 200), (3, 1, 32, 32, 12, 200, 200)]
for k, stride, cin, cout, n, H, W in CASES:
    x = Buf(torch.randn(n, H, W, cin, device=DEV), n, H, W, cin)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), DEV, stride=stride, act=native.ACT_RELU)
    out = Buf.alloc(n, H, W, cout, DEV)
    for _ in range(3):
        op([x], out)
    torch.cuda.synchronize()
    for persistent in os.environ.get('TRACE_PERSISTENT', '0,1').split(','):
        os.environ['FIERY_CONV_PERSISTENT'] = persistent
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            op([x], out)
        e.record()
        torch.cuda.synchronize()
        plain_us = s.elapsed_time(e) * 100
        probe = torch.zeros(8, dtype=torch.int64, device=DEV)
        trace = torch.zeros(8 * 40000, dtype=torch.int64, device=DEV)
        probe[7] = trace.data_ptr()
        os.environ['FIERY_CONV_CLKPROBE'] = hex(probe.data_ptr())
        op([x], out)
        torch.cuda.synchronize()
        s.record()
        op([x], out)
        e.record()
        torch.cuda.synchronize()
        del os.environ['FIERY_CONV_CLKPROBE']
        t = trace.view(-1, 8).cpu().numpy()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        us = (t[:, :4] - t0) / 100.0
        hw, xcc = t[:, 4], t[:, 5] & 15
        cu = (hw >> 8) & 15
        se = (hw >> 13) & 7
        sh = (hw >> 12) & 1
        simd = (hw >> 4) & 3
        place = xcc * 1000 + se * 100 + sh * 50 + cu            # one id per CU
        def q(a):
            return ' / '.join(f'{v:6.1f}' for v in np.percentile(a, [0, 10, 50, 90, 100]))
        print(f'conv k{k} {cin}->{cout} n={n} {H}x{W} persistent={persistent}: {plain_us:.1f} us per launch, traced launch {s.elapsed_time(e) * 1e3:.1f} us, '
              f'{len(us)} tiles on {len(np.unique(place))} CUs', flush=True)
        print(f'    entry {q(us[:, 0])} | set-up {q(us[:, 1] - us[:, 0])} | K loop {q(us[:, 2] - us[:, 1])} | epilogue {q(us[:, 3] - us[:, 2])} | '
              f'end {q(us[:, 3])}   (us: min / p10 / median / p90 / max)', flush=True)
        if (t[:, 6] > 0).all():                                  # (epilogue stamps: staged + barrier, scale / shift arrived)
            e6, e7 = (t[:, 6] - t0) / 100.0, (t[:, 7] - t0) / 100.0
            print(f'    epilogue parts: stage + barrier {q(e6 - us[:, 2])} | scale / shift arrive {q(e7 - e6)} | rows {q(us[:, 3] - e7)}', flush=True)
        per_cu = np.array([np.sum(place == c) for c in np.unique(place)])
        busy = np.array([(us[place == c, 3].max() - us[place == c, 0].min()) for c in np.unique(place)])
        print(f'    tiles per CU {per_cu.min()}..{per_cu.max()} (mean {per_cu.mean():.2f}); a CU is busy {q(busy)} us; '
              f'tiles of the busiest-count CUs end at {q(np.array([us[place == c, 3].max() for c in np.unique(place)[per_cu == per_cu.max()]]))}', flush=True)
        if out_dir:
            np.save(os.path.join(out_dir, f'conv_trace_k{k}_{cin}_{cout}_n{n}_{H}x{W}_p{persistent}.npy'), t)
    os.environ.pop('FIERY_CONV_PERSISTENT', None)
