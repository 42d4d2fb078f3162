#!/usr/bin/env python
"""Round 5 tuning aid: per-workgroup timeline of the Winograd kernel (library built with -DW_TRACE=1): how long the prologue, the K loop and
the epilogue of a workgroup take on the wall clock, and how much of a CU's time has 0 / 1 / 2 workgroups inside their K loops."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
a = [int(v) for v in sys.argv[1:]]
cin, cout, n, H, W = (a + [128, 128, 3, 200, 200])[:5] if len(a) >= 5 else (128, 128, 3, 200, 200)
tiles = n * ((H + 1) // 2) * ((W + 1) // 2)
grid = ((tiles + 31) // 32) * (cout // 64)
trace = torch.zeros(grid * 8, dtype=torch.int64, device='cuda')
os.environ['FIERY_WINOGRAD_TRACE'] = hex(trace.data_ptr())
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402
lib = native.get()
x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=True)
op.force_form = 'wino'
out = Buf.alloc(n, H, W, cout, 'cuda')
for _ in range(5):
    op([x], out)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(grid, 8).astype(np.int64)
t0 = t[:, 0].min()
ent, pro, kl, end = (t[:, i] - t0 for i in range(4))
us = lambda v: v / 100.0
print(f'{cin}->{cout} {n}x{H}x{W}: {grid} workgroups; launch {us(end.max()):.1f} us on the wall clock')
for name, d in (('prologue (entry -> first barrier)', pro - ent), ('K loop', kl - pro), ('epilogue', end - kl), ('whole workgroup', end - ent)):
    d = us(d)
    print(f'  {name:36s} mean {d.mean():6.2f} us   p10 {np.percentile(d, 10):6.2f}   median {np.median(d):6.2f}   p90 {np.percentile(d, 90):6.2f}')
# CU occupancy by phase: CU id = (xcc, se, cu) from HW_ID
hw, xcc = t[:, 4], t[:, 5] & 0xF
cu_key = (xcc << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xF)          # se_id bits 15:13, cu_id bits 11:8
span = end.max()
acc = np.zeros(4)
for key in np.unique(cu_key):
    sel = cu_key == key
    ev = []
    for a_, b_, c_ in zip(pro[sel], kl[sel], end[sel]):
        ev += [(a_, 1), (b_, -1)]
    ev.sort()
    cur, last = 0, 0
    for tt, dlt in ev:
        acc[min(cur, 3)] += tt - last
        cur += dlt
        last = tt
    acc[0] += span - last
tot = acc.sum()
print('  time of a CU with 0 / 1 / 2 / 3+ workgroups inside their K loops: ' + ' / '.join(f'{100 * v / tot:.1f} %' for v in acc),
      f'({len(np.unique(cu_key))} CUs seen)')
ideal = (cin // 16) * 64 * 64 / 2.4e3     # us of MFMA time of one wavefront's K loop at 2.4 GHz, alone on its SIMD
print(f'  a K loop is {(cin // 16) * 64} MFMAs per wavefront = {ideal:.2f} us alone on a SIMD at 2.4 GHz; two wavefronts share a SIMD')
