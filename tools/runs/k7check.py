import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fiery_amd import native
from fiery_amd.ops import Buf, ConvOp, identity_chan_map
DEV = 'cuda:0'
lib = native.get()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
cases = [(7, 2, 64, 64, 15, 200, 200), (3, 2, 64, 128, 15, 100, 100), (3, 1, 128, 128, 3, 200, 200), (7, 2, 64, 64, 2, 200, 200), (5, 2, 64, 64, 15, 200, 200), (7, 1, 64, 64, 4, 200, 200)]
for k, stride, cin, cout, n, H, W in cases:
    torch.manual_seed(0)
    x = Buf(torch.randn(n, H, W, cin, device=DEV), n, H, W, cin)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    ref = None
    for tm in (64, 128):
        for al in (0, 1):
            os.environ['FIERY_CONV_TILE_M'] = str(tm)
            os.environ['FIERY_CONV_ALIGNED'] = str(al)
            op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), DEV, stride=stride, act=native.ACT_RELU)
            ho, wo = op.out_hw(H, W)
            out = Buf.alloc(n, ho, wo, cout, DEV)
            us = timed(lambda: op([x], out))
            o = out.nhwc().clone()
            if ref is None: ref = o
            print(f'k{k} s{stride} {cin}->{cout} n={n} {H}x{W} tile_m={tm} aligned={al}: {us:8.1f} us  max|diff| vs first = {(o - ref).abs().max().item():.3e}', flush=True)
