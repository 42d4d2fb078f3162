#!/usr/bin/env python
"""Round 6: variants of the pooling op on one box, alternating, in one process - the frame-group pipeline (prepass of group
i + 1 under the rows of group i, `native.voxel_pool_grouped`) and the 384-thread form next to the single launch pair.
  python tools/runs/r6_pool_ab.py [preset] "GROUPS=4+5,ENV=VAL" "ENV=VAL" ...      ("-" = defaults; GROUPS is read here)
Each variant is timed three ways: back-to-back eager ops (median), cold (512 MB copied between ops), replays of a captured
graph of the op (what the engine serves)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.synthetic import make_inputs                    # noqa: E402

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
from fiery_amd.model import calculate_birds_eye_view_parameters       # noqa: E402
res, start, dim = (t.numpy() for t in calculate_birds_eye_view_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND))
X, Y = int(dim[0]), int(dim[1])
grid = native.make_grid((start - res / np.float32(2)).astype(np.float32), res, dim)
H_img, W_img = cfg.IMAGE.FINAL_DIM
depth = torch.arange(*cfg.LIFT.D_BOUND, dtype=torch.float)
fh_, fw_ = H_img // 8, W_img // 8
frustum = torch.stack((torch.linspace(0, W_img - 1, fw_).view(1, 1, fw_).expand(len(depth), fh_, fw_),
                       torch.linspace(0, H_img - 1, fh_).view(1, fh_, 1).expand(len(depth), fh_, fw_),
                       depth.view(-1, 1, 1).expand(len(depth), fh_, fw_)), -1).contiguous().to(DEV)
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
out = torch.empty(frames, 64, X, Y, device=DEV)
junk_a = torch.empty(512 * 1024 * 1024 // 4, device=DEV)
junk_b = torch.empty_like(junk_a)
ws_clean = lib.pool_workspace(frames, n_cam, D, fh, fw, DEV, grid, 0, 0, zeroed=True)
sides = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
GROUPS = None
EXTRA_FLAGS = 0


def op():
    if GROUPS:
        lib.voxel_pool_grouped(x, strides, geo, frames, n_cam, D, fh, fw, 64, grid, GROUPS, sides, out=out, workspace=ws_clean,
                               flags=(0 if EXTRA_FLAGS >> 30 else native.POOL_WORKSPACE_CLEAN) | (EXTRA_FLAGS & 0xffff))
    else:
        lib.voxel_pool(x, strides, geo, frames, n_cam, D, fh, fw, 64, grid, out=out, workspace=ws_clean, flags=(0 if EXTRA_FLAGS >> 30 else native.POOL_WORKSPACE_CLEAN) | (EXTRA_FLAGS & 0xffff))


def setenv(v):
    global GROUPS, EXTRA_FLAGS
    touched = []
    GROUPS = None
    EXTRA_FLAGS = 0
    if v != '-':
        for kv in v.split(','):
            k, val = kv.split('=')
            if k == 'GROUPS':
                GROUPS = tuple(int(t) for t in val.split('+'))
                continue
            if k == 'NO_RANKS':
                EXTRA_FLAGS |= native.POOL_NO_RANKS if int(val) else 0
                continue
            if k == 'NOCLEAN':                       # without POOL_WORKSPACE_CLEAN: the call clears its workspace with a memset dispatch
                EXTRA_FLAGS |= 1 << 30
                continue
            os.environ[k] = val
            touched.append(k)
    return touched


ref = None
print(f'{preset}: {frames} frames x {n_cam} cams, {X}x{Y}, algorithmic {algo / 1e6:.1f} MB, kept {n_kept / n_pts:.3f}', flush=True)
for r in range(rounds):
    for v in variants:
        touched = setenv(v)
        out.fill_(float('nan'))
        try:
            op()
        except Exception as exc:                                        # noqa: BLE001  (an older library build without the flag)
            print(f'[{r}] {v:44s} not run: {exc}', flush=True)
            for k in touched:
                del os.environ[k]
            continue
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
        warm = per[len(per) // 2]
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
        cold_us = sorted(cold)[len(cold) // 2]
        if os.environ.get('PRE_MODES') == '1':
            # what the op's bracket reads behind different predecessors (bench.py brackets it behind a 25 ms GPU-side sleep)
            mm_a = torch.randn(4096, 4096, device=DEV)
            for mode in ('sleep', 'sleep+geometry', 'matmul', 'matmul+geometry', 'pool', 'sleep+pool'):
                vals = []
                for _ in range(10):
                    if 'sleep' in mode:
                        torch.cuda._sleep(20_000_000)
                    if 'matmul' in mode:
                        for _ in range(4):
                            mm_a @ mm_a
                    if 'pool' in mode:
                        op()
                    if 'geometry' in mode:
                        lib.lift_geometry(frustum, cam)
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record()
                    op()
                    e_.record()
                    torch.cuda.synchronize()
                    vals.append(s_.elapsed_time(e_) * 1e3)
                print(f'      behind {mode:18s} median {sorted(vals)[5]:6.1f} us  min {min(vals):6.1f}  max {max(vals):6.1f}', flush=True)
        # graph replays: ten ops per graph, the graph replayed `reps` times
        cap = torch.cuda.Stream(DEV)
        cap.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(10):
                op()
        g.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps):
            g.replay()
        e_.record()
        torch.cuda.synchronize()
        graph_us = s_.elapsed_time(e_) * 1e3 / reps / 10
        err2 = (out - ref).abs().max().item()
        print(f'[{r}] {v:44s} warm {warm:6.1f} us ({algo / warm / 1e3 / 8000:.1%})  cold {cold_us:6.1f} us ({algo / cold_us / 1e3 / 8000:.1%})  '
              f'graph {graph_us:6.1f} us ({algo / graph_us / 1e3 / 8000:.1%})  max|d| vs first {max(err, err2):.2e}', flush=True)
        del g
        for k in touched:
            del os.environ[k]
