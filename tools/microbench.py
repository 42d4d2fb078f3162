#!/usr/bin/env python
"""Kernel microbenchmarks for tuning (GPU box): the voxel-pooling op at baseline.yml size and the convolution
shapes that dominate the step.  Prints one line per case; run under rocprofv3 (--kernel-trace / --pmc) for counters.
  python tools/microbench.py [probe] [pool] [conv] [--reps N]      (env: POOL_TILES=a,b  POOL_FRAMES=9,12)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fiery_amd import native                                   # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402
from fiery_amd.synthetic import make_inputs                    # noqa: E402
from oracle import lift_splat as ls                            # noqa: E402

DEV = 'cuda:0'


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps      # us


def bench_probe(reps):
    """Plain streaming reads of a 1.1 GB tensor: the practical HBM read ceiling of this box."""
    import ctypes as C
    so = os.path.join(ROOT, 'tools', 'probe', 'libhbm_probe.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                               os.path.join(ROOT, 'tools', 'probe', 'hbm_probe.hip'), '-o', so])
    probe = C.CDLL(so)
    probe.probe_read.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    x = torch.randn(9 * 6 * 64 * 48 * 28 * 60 // 4, 4, device=DEV)
    sink = torch.zeros(4, device=DEV)
    nbytes = x.numel() * 4
    stream = torch.cuda.current_stream().cuda_stream
    for blocks, threads, unroll in ((1152, 512, 16), (1152, 512, 28), (1152, 512, 8), (2304, 256, 16), (576, 1024, 16),
                                    (4608, 512, 16), (512, 512, 16), (1024, 512, 28)):
        us = timed(lambda: probe.probe_read(x.data_ptr(), nbytes, blocks, threads, unroll, 3, sink.data_ptr(), stream), reps)
        print(f'probe pooling pattern (28 rows x 16 B per lane) blocks={blocks:5d} threads={threads:4d} rows in flight={unroll:2d}: '
              f'{us:7.1f} us -> {nbytes / us / 1e3:7.1f} GB/s', flush=True)
    for mode, name in ((0, 'grid-stride'), (1, 'grid-stride nontemporal'), (2, 'slab per workgroup')):
        for blocks, threads, unroll in ((2048, 256, 4), (4096, 256, 8), (1024, 512, 8), (512, 1024, 8), (2048, 512, 16),
                                        (8192, 256, 4), (1152, 512, 16), (1024, 512, 16)):
            us = timed(lambda: probe.probe_read(x.data_ptr(), nbytes, blocks, threads, unroll, mode, sink.data_ptr(), stream), reps)
            print(f'probe {name:24s} blocks={blocks:5d} threads={threads:4d} unroll={unroll:2d}: {us:7.1f} us -> '
                  f'{nbytes / us / 1e3:7.1f} GB/s', flush=True)


def bench_pool(lib, reps, frames=9, tiles=(0,)):
    cfg = get_preset_cfg(os.environ.get('POOL_PRESET', 'baseline.yml'))
    res, start, dim = ls.bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
    X, Y = int(dim[0]), int(dim[1])
    grid = native.make_grid((start - res / np.float32(2)).astype(np.float32), res, dim)
    frustum = torch.from_numpy(ls.create_frustum(cfg.IMAGE.FINAL_DIM, 8, cfg.LIFT.D_BOUND)).to(DEV)
    D, fh, fw = frustum.shape[:3]
    _, K, E, _ = make_inputs(frames // 3, 3, 6, with_image=False)
    cam = lib.camera_matrices(K.view(-1, 3, 3).to(DEV), E.view(-1, 4, 4).to(DEV))
    geo = lib.lift_geometry(frustum, cam).view(frames, 6, D, fh, fw, 3)
    x = torch.randn(frames, 6, 64, D, fh, fw, device=DEV)
    st = x.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    # platform calibration on the same tensor: what plain streaming kernels reach here
    us = timed(lambda: x.sum(), reps)
    print(f'calib torch.sum over x ({x.numel() * 4 / 1e6:.0f} MB): {us:8.1f} us -> {x.numel() * 4 / us / 1e3:7.1f} GB/s read', flush=True)
    y = torch.empty_like(x)
    us = timed(lambda: y.copy_(x), reps)
    print(f'calib copy x -> y: {us:8.1f} us -> {2 * x.numel() * 4 / us / 1e3:7.1f} GB/s read+write', flush=True)
    del y
    rank, _ = lib.voxel_index(geo, grid, want_idx=False)
    n_kept = int((rank >= 0).sum())
    n_pts = rank.numel()
    algo = 4.0 * 64 * n_kept + 12.0 * n_pts + 4.0 * 64 * frames * X * Y
    for tile in tiles:
        ws = lib.pool_workspace(frames, 6, D, fh, fw, DEV, grid, tile, 0)
        out = torch.empty(frames, 64, X, Y, device=DEV)
        us = timed(lambda: lib.voxel_pool(x, strides, geo, frames, 6, D, fh, fw, 64, grid, out=out, workspace=ws,
                                          tile_voxels=tile), reps)
        # the same op with the caches (L2, 256 MB Infinity Cache) flushed by an unrelated 2 GB copy before every call -
        # what the op sees inside a step
        junk_a = torch.empty(512 * 1024 * 1024 // 4, device=DEV)
        junk_b = torch.empty_like(junk_a)
        cold = []
        for _ in range(reps):
            junk_b.copy_(junk_a)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            lib.voxel_pool(x, strides, geo, frames, 6, D, fh, fw, 64, grid, out=out, workspace=ws, tile_voxels=tile)
            e_.record()
            torch.cuda.synchronize()
            cold.append(s_.elapsed_time(e_) * 1e3)
        del junk_a, junk_b
        print(f'pool frames={frames} cold caches: {sum(cold) / len(cold):8.1f} us/op (min {min(cold):.1f})', flush=True)
        if '--probe' in sys.argv:
            probe = torch.zeros(8, dtype=torch.int64, device=DEV)
            os.environ['FIERY_POOL_PROBE'] = hex(probe.data_ptr())
            us_p = timed(lambda: lib.voxel_pool(x, strides, geo, frames, 6, D, fh, fw, 64, grid, out=out, workspace=ws,
                                                tile_voxels=tile), reps)
            del os.environ['FIERY_POOL_PROBE']
            tot, clr, ld, acc, mrg, fin, nwg, nit = (int(v) for v in probe.cpu())
            nwg = max(nwg, 1)
            print(f'  probe ({us_p:.1f} us/op): per sampled workgroup (wave 0) {tot / nwg:9.0f} cycles = clear {clr / nwg:7.0f} + '
                  f'row loads issue/wait {ld / nwg:8.0f} + rest of the item loop {acc / nwg:8.0f} + merge/LDS adds {mrg / nwg:8.0f} + '
                  f'barrier/write-out {fin / nwg:7.0f}; {nit / nwg:.1f} items per thread', flush=True)
        print(f'pool frames={frames} tile={tile or "default"}: {us:8.1f} us/op  {us / frames:6.1f} us/frame  '
              f'algorithmic {algo / 1e6:.1f} MB -> {algo / us / 1e3:7.1f} GB/s ({algo / us / 1e3 / 8000:.1%} of 8 TB/s); kept {n_kept / n_pts:.3f}',
              flush=True)


    # backward (training): grad_x = gather of grad_out through the ranks the forward call left in the workspace.
    # algorithmic bytes: every grad_x element written once + the ranks + the dense gradient read once
    gx = torch.empty_like(x).permute(0, 1, 3, 4, 5, 2)
    gout = torch.randn(frames, 64, 200, 200, device=DEV)
    rk = ws[:n_pts]
    us = timed(lambda: lib.voxel_pool_bwd(gout, rk, frames, 6, D, fh, fw, 64, gx), reps)
    algo_b = 4.0 * 64 * n_pts + 4.0 * n_pts + 4.0 * 64 * frames * 200 * 200
    print(f'pool bwd frames={frames}: {us:8.1f} us/op  {us / frames:6.1f} us/frame  algorithmic {algo_b / 1e6:.1f} MB -> '
          f'{algo_b / us / 1e3:7.1f} GB/s ({algo_b / us / 1e3 / 8000:.1%} of 8 TB/s)', flush=True)
    prob = torch.rand(frames, 6, D, fh, fw, device=DEV)
    feat = torch.randn(frames, 6, 64, fh, fw, device=DEV)
    us = timed(lambda: lib.lift_splat_bwd(gout, rk, prob, feat, frames, 6, D, fh, fw, 64), reps)
    print(f'lift-splat bwd frames={frames}: {us:8.1f} us/op  {us / frames:6.1f} us/frame', flush=True)


CONV_CASES = [  # (k, stride, cin, cout, n_img, H, W, residual)
    (3, 1, 128, 128, 3, 200, 200, False),
    (3, 1, 128, 64, 3, 200, 200, False),
    (3, 1, 64, 256, 15, 200, 200, False),
    (3, 1, 32, 32, 12, 200, 200, False),
    (1, 1, 32, 64, 12, 200, 200, True),
    (1, 1, 64, 32, 12, 200, 200, False),
    (7, 2, 64, 64, 15, 200, 200, False),
    # tile-count quantisation probes: 147456 = 768 * 192 pixel rows (whole rounds of 64-row tiles at 3 per CU)
    (3, 1, 128, 128, 3, 192, 256, False),
    (3, 1, 128, 128, 6, 192, 256, False),
    (3, 1, 128, 64, 3, 192, 256, False),
    # partial rounds: 768 / 384 / 339 / 1536 tiles of 64 rows
    (3, 1, 128, 128, 1, 192, 256, False),
    (3, 1, 128, 128, 1, 192, 128, False),
    (3, 1, 128, 128, 1, 96, 226, False),
    (3, 1, 128, 128, 2, 192, 256, False),
]


def bench_conv(lib, reps):
    for k, stride, cin, cout, n, H, W, residual in CONV_CASES:
        x = Buf(torch.randn(n, H, W, cin, device=DEV), n, H, W, cin)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        prec = native.PRECISION_BF16 if os.environ.get('CONV_PRECISION') == 'bf16' else native.PRECISION_F32
        op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), DEV, stride=stride,
                    act=native.ACT_RELU, precision=prec)
        ho, wo = op.out_hw(H, W)
        out = Buf.alloc(n, ho, wo, cout, DEV)
        res = Buf(torch.randn(n, ho, wo, cout, device=DEV), n, ho, wo, cout) if residual else None
        us = timed(lambda: op([x], out, res=res), reps)
        flops = 2.0 * n * ho * wo * cin * k * k * cout
        clk = ''
        if '--clk' in sys.argv:                  # effective shader clock under this launch (instrumented kernel variant)
            probe = torch.zeros(8, dtype=torch.int64, device=DEV)
            os.environ['FIERY_CONV_CLKPROBE'] = hex(probe.data_ptr())
            us_p = timed(lambda: op([x], out, res=res), reps)
            del os.environ['FIERY_CONV_CLKPROBE']
            cyc, ticks, pa, pb, pc, pd, nwg = (int(v) for v in probe.cpu()[:7])
            mhz = 100.0 * cyc / max(ticks, 1)
            nwg = max(nwg, 1)
            clk = (f'\n      shader clock {mhz:6.0f} MHz; probe run {us_p:.1f} us; per sampled workgroup (wave 0): K loop {cyc / nwg:9.0f} cycles = '
                   f'load issue {pa / nwg:8.0f} + LDS reads/MFMA {pb / nwg:8.0f} + load wait/LDS store {pc / nwg:8.0f} + barrier {pd / nwg:8.0f}')
        print(f'conv k{k} s{stride} {cin:3d}->{cout:3d} n={n:2d} {H}x{W} [tile {list(op._tile_m.values())}]: {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s '
              f'({flops / us / 1e6 / 157.3:.1%} of fp32 MFMA peak){clk}', flush=True)


def bench_chains(lib, reps):
    """Does per-sample stream parallelism hide the partly filled last round of workgroups?  A GRU-like dependent
    chain (gates 128->128 then tilde 128->64, 4 steps) for 3 samples: batched on one stream vs one stream per sample."""
    H = W = 200
    cases = [(128, 128), (128, 64)]
    ops = []
    for cin, cout in cases:
        w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        ops.append(ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), DEV,
                          act=native.ACT_RELU))

    def make(n):
        x = Buf(torch.randn(n, H, W, 128, device=DEV), n, H, W, 128)
        return x, Buf.alloc(n, H, W, 128, DEV), Buf.alloc(n, H, W, 64, DEV)

    def chain(bufs, steps=4):
        x, a, b = bufs
        for _ in range(steps):
            ops[0]([x], a)
            ops[1]([a], b)

    batched = make(3)
    us = timed(lambda: chain(batched), reps)
    flops = 4 * 3 * 2.0 * H * W * 128 * 9 * (128 + 64)
    print(f'chain batched n=3, 1 stream : {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s', flush=True)
    singles = [make(1) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]

    def fan():
        cur = torch.cuda.current_stream()
        for s_, bufs in zip(streams, singles):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                chain(bufs)
        for s_ in streams:
            cur.wait_stream(s_)

    us = timed(fan, reps)
    print(f'chain 3 x n=1, 3 streams   : {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s', flush=True)
    g = torch.cuda.CUDAGraph()
    fan()
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            fan()
        us = timed(g.replay, reps)
        print(f'chain 3 x n=1, hipGraph    : {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s', flush=True)
    except Exception as e:                                       # noqa: BLE001
        print('graph capture failed:', repr(e)[:300], flush=True)
    g2 = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g2):
            chain(batched)
        us = timed(g2.replay, reps)
        print(f'chain batched n=3, hipGraph : {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s', flush=True)
    except Exception as e:                                       # noqa: BLE001
        print('graph capture failed:', repr(e)[:300], flush=True)


if __name__ == '__main__':
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
    lib = native.get()
    what = [a for a in sys.argv[1:] if a in ('pool', 'conv', 'probe', 'chains')] or ['pool', 'conv']
    if 'probe' in what:
        bench_probe(reps)
    if 'pool' in what:
        tiles = tuple(int(t) for t in os.environ.get('POOL_TILES', '0').split(','))
        for frames in (int(f) for f in os.environ.get('POOL_FRAMES', '9').split(',')):
            bench_pool(lib, reps, frames=frames, tiles=tiles)
    if 'conv' in what:
        bench_conv(lib, reps)
    if 'chains' in what:
        bench_chains(lib, reps)
