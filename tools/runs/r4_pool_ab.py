#!/usr/bin/env python
"""Round 4: variants of the pooling op on one box, alternating, in one process (the library reads its tuning switches per call).
  python tools/runs/r4_pool_ab.py [preset] "ENV=VAL,ENV=VAL" "ENV=VAL" ...      ("-" = defaults)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.synthetic import make_inputs                    # noqa: E402
from oracle import lift_splat as ls                            # noqa: E402

DEV = 'cuda:0'
args = sys.argv[1:]
preset = 'baseline.yml'
if args and args[0].endswith('.yml'):
    preset, args = args[0], args[1:]
variants = args or ['-']
reps = int(os.environ.get('REPS', '30'))
rounds = int(os.environ.get('ROUNDS', '2'))
frames = int(os.environ.get('FRAMES', '9'))
n_cam = int(os.environ.get('CAMS', '6'))

lib = native.get()
cfg = get_preset_cfg(preset)
res, start, dim = ls.bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
X, Y = int(dim[0]), int(dim[1])
grid = native.make_grid((start - res / np.float32(2)).astype(np.float32), res, dim)
frustum = torch.from_numpy(ls.create_frustum(cfg.IMAGE.FINAL_DIM, 8, cfg.LIFT.D_BOUND)).to(DEV)
D, fh, fw = frustum.shape[:3]
_, K, E, _ = make_inputs(frames // 3, 3, n_cam, with_image=False)
cam = lib.camera_matrices(K.view(-1, 3, 3).to(DEV), E.view(-1, 4, 4).to(DEV))
geo = lib.lift_geometry(frustum, cam).view(frames, n_cam, D, fh, fw, 3)
x = torch.randn(frames, n_cam, 64, D, fh, fw, device=DEV)
st = x.stride()
strides = (st[0], st[1], st[3], st[4], st[5], st[2])
rank, _ = lib.voxel_index(geo, grid, want_idx=False)
n_kept, n_pts = int((rank >= 0).sum()), rank.numel()
algo = 4.0 * 64 * n_kept + 12.0 * n_pts + 4.0 * 64 * frames * X * Y
ws = lib.pool_workspace(frames, n_cam, D, fh, fw, DEV, grid, 0, 0)
out = torch.empty(frames, 64, X, Y, device=DEV)
junk_a = torch.empty(512 * 1024 * 1024 // 4, device=DEV)
junk_b = torch.empty_like(junk_a)


ws_clean = lib.pool_workspace(frames, n_cam, D, fh, fw, DEV, grid, 0, 0, zeroed=True)


def op():
    if os.environ.get('PY_CLEAN') == '1':          # the engine's way: zero-filled once, then POOL_WORKSPACE_CLEAN (no memset dispatch)
        lib.voxel_pool(x, strides, geo, frames, n_cam, D, fh, fw, 64, grid, out=out, workspace=ws_clean, flags=native.POOL_WORKSPACE_CLEAN)
    else:
        lib.voxel_pool(x, strides, geo, frames, n_cam, D, fh, fw, 64, grid, out=out, workspace=ws)


def setenv(v):
    touched = []
    if v != '-':
        for kv in v.split(','):
            k, val = kv.split('=')
            os.environ[k] = val
            touched.append(k)
    return touched


ref = None
print(f'{preset}: {frames} frames x {n_cam} cams, {X}x{Y}, algorithmic {algo / 1e6:.1f} MB, kept {n_kept / n_pts:.3f}', flush=True)
for r in range(rounds):
    for v in variants:
        touched = setenv(v)
        op()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        err = (out - ref).abs().max().item()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        evs[0].record()
        for i in range(reps):
            op()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(reps))
        warm = per[len(per) // 2]                                       # median of the back-to-back ops
        cold = []
        for _ in range(reps):
            junk_b.copy_(junk_a)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            op()
            e_.record()
            torch.cuda.synchronize()
            cold.append(s_.elapsed_time(e_) * 1e3)
        cold_us = sorted(cold)[len(cold) // 2]
        print(f'[{r}] {v:60s} warm {warm:7.1f} us ({algo / warm / 1e3 / 8000:.1%})  cold {cold_us:7.1f} us ({algo / cold_us / 1e3 / 8000:.1%}, min {min(cold):.1f})  '
              f'max|d| vs first {err:.2e}', flush=True)
        if os.environ.get('TRACE') == '1' and r == rounds - 1:
            # timeline of one launch: the 100 MHz wall clock at every item's phases (FIERY_POOL_TRACE)
            tr = torch.zeros(4 * 4096, dtype=torch.int64, device=DEV)
            os.environ['FIERY_POOL_TRACE'] = hex(tr.data_ptr())
            junk_b.copy_(junk_a)
            op()
            torch.cuda.synchronize()
            del os.environ['FIERY_POOL_TRACE']
            t = tr.view(-1, 4).cpu().numpy()
            if os.environ.get('TRACE_DUMP'):
                np.save(os.path.join(os.environ['TRACE_DUMP'], 'trace_' + ''.join(ch if ch.isalnum() else '_' for ch in v) + '.npy'), t)
            t = t[t[:, 0] > 0]
            t0 = t[:, 0].min()
            us = (t - t0) / 100.0
            n_main = min(len(us), 64 * frames - (64 * frames) % 512) if len(us) > 512 else len(us)
            def q(a):
                return ' / '.join(f'{v:6.1f}' for v in np.percentile(a, [0, 10, 50, 90, 100]))
            for name, sel in (('whole units', slice(0, n_main)), ('tail parts', slice(n_main, None))):
                u = us[sel]
                if len(u) == 0:
                    continue
                print(f'      {name:11s} n={len(u):4d}  start {q(u[:, 0])} | set-up {q(u[:, 1] - u[:, 0])} | rows {q(u[:, 2] - u[:, 1])} | '
                      f'write-out {q(u[:, 3] - u[:, 2])} | end {q(u[:, 3])}   (us: min / p10 / median / p90 / max)', flush=True)
        for k in touched:
            del os.environ[k]
