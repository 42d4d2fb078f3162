"""Round 3: what each phase of the Bottleneck tail launch costs (12 x 200 x 200 images, the future-prediction res blocks).
Variants of the SAME kernel: 3x3 32->32 alone; + chained 1x1 32->64; + residual; + the next block's 64->32 (the real launch)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

DEV = 'cuda:0'
RELU = native.ACT_RELU


def timed(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    lib = native.get()
    n, H, W = int(os.environ.get('TAIL_IMAGES', '12')), 200, 200
    g = torch.Generator().manual_seed(0)
    w3 = torch.randn(32, 32, 3, 3, generator=g) / 17
    wu = torch.randn(64, 32, 1, 1, generator=g) / 6
    wd = torch.randn(32, 64, 1, 1, generator=g) / 8
    one32, zero32 = torch.ones(32), torch.zeros(32)
    one64, zero64 = torch.ones(64), torch.zeros(64)
    t1 = Buf(torch.randn(n, H, W, 32, device=DEV), n, H, W, 32)
    x = Buf(torch.randn(n, H, W, 64, device=DEV), n, H, W, 64)
    out32, out64, nxt = Buf.alloc(n, H, W, 32, DEV), Buf.alloc(n, H, W, 64, DEV), Buf.alloc(n, H, W, 32, DEV)
    px = n * H * W
    plain = ConvOp(lib, w3, identity_chan_map(32), (4, 0), one32, zero32, DEV, act=RELU)
    tail = ConvOp(lib, w3, identity_chan_map(32), (4, 0), one32, zero32, DEV, act=RELU).chain_pointwise(wu, one64, zero64, RELU)
    full = tail.chain_next(wd, one32, zero32, RELU)
    up = ConvOp(lib, wu, identity_chan_map(32), (4, 0), one64, zero64, DEV, act=RELU)
    down = ConvOp(lib, wd, identity_chan_map(64), (8, 0), one32, zero32, DEV, act=RELU)
    cases = [
        ('3x3 32->32 (+BN+ReLU), 32 channels stored', lambda: plain([t1], out32), 2.0 * px * 9 * 32 * 32),
        ('+ chained 1x1 32->64, no residual', lambda: tail([t1], out64), 2.0 * px * (9 * 32 * 32 + 32 * 64)),
        ('+ residual', lambda: tail([t1], out64, res=x), 2.0 * px * (9 * 32 * 32 + 32 * 64)),
        ('+ next block\'s 1x1 64->32 (the launch of the step)', lambda: full([t1], out64, res=x, out3=nxt), 2.0 * px * (9 * 32 * 32 + 32 * 64 + 64 * 32)),
        ('separate launch: 1x1 32->64 + residual', lambda: up([out32], out64, res=x), 2.0 * px * 32 * 64),
        ('separate launch: 1x1 64->32', lambda: down([out64], nxt), 2.0 * px * 64 * 32),
    ]
    for name, fn, flops in cases:
        us = timed(fn)
        extra = ''
        if '--clk' in sys.argv:              # FIERY_CONV_TUNING=1 builds: cycle stamps of sampled workgroups (wave 0)
            probe = torch.zeros(8, dtype=torch.int64, device=DEV)
            os.environ['FIERY_CONV_CLKPROBE'] = hex(probe.data_ptr())
            us_p = timed(fn, 5)
            del os.environ['FIERY_CONV_CLKPROBE']
            cyc, ticks, pro, epi, _, _, nwg = (int(v) for v in probe.cpu()[:7])
            nwg = max(nwg, 1)
            extra = (f'   [probe run {us_p:.1f} us, {100.0 * cyc / max(ticks, 1):.0f} MHz; per workgroup: prologue {pro / nwg:7.0f} + K loop '
                     f'{cyc / nwg:7.0f} + epilogue {epi / nwg:7.0f} cycles]')
        print(f'{name:58s} {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s{extra}', flush=True)


if __name__ == '__main__':
    main()
