"""TEST INFRASTRUCTURE: run one kernel family of the CPU-simulated library on operands that end (or begin) exactly at an
inaccessible page, so that a read or write outside an operand faults instead of passing unnoticed.

    python tests/sim/guard_check.py <case> <end|begin>

Buffer loads are bounds-checked by their descriptors in the simulator as on the hardware; what this catches are plain
pointer accesses (16-byte row loads, float4 stores, scalar tails) that run past a tensor.  A fault kills the process (the
test that launches this script reports the faulthandler trace)."""
import ctypes
import faulthandler
import mmap
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_keep = []


def guarded(src, where):
    """A copy of `src` whose last byte (where='end') or first byte ('begin') sits next to a PROT_NONE page."""
    nbytes = src.numel() * src.element_size()
    body = (nbytes + PAGE - 1) // PAGE * PAGE
    region = mmap.mmap(-1, body + 2 * PAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(region))
    for guard in (base, base + PAGE + body):
        if _libc.mprotect(ctypes.c_void_p(guard), PAGE, 0) != 0:
            raise OSError(ctypes.get_errno(), 'mprotect')
    offset = PAGE + (body - nbytes if where == 'end' else 0)
    assert (base + offset) % 16 == 0 or where == 'end' and nbytes % 16, 'operands are 16-byte aligned in the product too'
    buf = (ctypes.c_char * nbytes).from_address(base + offset)
    t = torch.frombuffer(buf, dtype=src.dtype).view(src.shape)
    t.copy_(src.contiguous())
    _keep.append(region)
    return t


def guard_allocations(where):
    """From here on `torch.empty / zeros / empty_like` hand out guarded host memory: the outputs and workspaces the bindings
    (fiery_amd/native.py) and the autograd Functions allocate are checked like the inputs."""
    real_empty, real_zeros, real_empty_like = torch.empty, torch.zeros, torch.empty_like

    def shape_of(args):
        return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else tuple(args)

    def make(shape, dtype, fill):
        src = real_zeros(shape, dtype=dtype) if fill else real_empty(shape, dtype=dtype)
        return guarded(src, where) if src.numel() else src

    def empty(*args, dtype=None, device=None, **kw):
        if (device is None or str(device) == 'cpu') and not kw and dtype in (torch.float32, torch.int32, torch.uint8, torch.bfloat16):
            return make(shape_of(args), dtype, False)
        return real_empty(*args, dtype=dtype, device=device, **kw)

    def zeros(*args, dtype=None, device=None, **kw):
        if (device is None or str(device) == 'cpu') and not kw and dtype in (torch.float32, torch.int32, torch.uint8):
            return make(shape_of(args), dtype, True)
        return real_zeros(*args, dtype=dtype, device=device, **kw)

    def empty_like(t, **kw):
        if not kw and t.device.type == 'cpu' and t.dtype in (torch.float32, torch.int32) and t.is_contiguous():
            return make(tuple(t.shape), t.dtype, False)
        return real_empty_like(t, **kw)

    torch.empty, torch.zeros, torch.empty_like = empty, zeros, empty_like


def main():
    faulthandler.enable()
    case, where = sys.argv[1], sys.argv[2]
    guard_allocations(where)
    from fiery_amd import native
    from fiery_amd import train_graph as tg
    from tests.sim.build_sim import build
    lib = native.Lib(build())
    g = torch.Generator().manual_seed(0)
    G = lambda *shape: guarded(torch.randn(*shape, generator=g), where)
    if case == 'conv':                                   # forward, input gradient, weight gradient (staged 3x3, 1x1, generic)
        for cin, cout, k, s, p, hw in ((16, 24, 3, 1, 1, (9, 11)), (32, 32, 3, 1, 1, (5, 58)), (64, 32, 1, 1, 0, (7, 13)), (8, 40, 3, 2, 1, (9, 12)),
                                       (35, 36, 3, 1, 1, (6, 25)), (11, 32, 7, 2, 3, (12, 12)), (40, 8, 1, 1, 0, (1, 1))):
            x = G(2, cin, *hw).contiguous(memory_format=torch.channels_last)
            x = guarded(x.permute(0, 2, 3, 1), where).permute(0, 3, 1, 2).requires_grad_()      # pixel-major memory, guarded
            w = (G(cout, cin, k, k) * 0.1).requires_grad_()
            y = tg.HipConv2d.apply(x, w, s, p, lib)
            gy = guarded(torch.randn(y.permute(0, 2, 3, 1).shape, generator=g), where).permute(0, 3, 1, 2)
            torch.autograd.grad(y, (x, w), gy)
    elif case == 'bn':
        for c, relu, training in ((64, True, True), (35, True, True), (6, False, True), (21, False, False), (256, True, True)):
            x = guarded(torch.randn(3, 5, 7, c, generator=g), where).permute(0, 3, 1, 2).requires_grad_()
            weight, bias = G(c).requires_grad_(), G(c).requires_grad_()
            rm, rv = G(c), guarded(torch.rand(c, generator=g) + 0.5, where)
            y = tg.HipBatchNormAct.apply(x, weight, bias, rm, rv, training, 0.1, 1e-5, relu, lib)
            gy = guarded(torch.randn(3, 5, 7, c, generator=g), where).permute(0, 3, 1, 2)
            torch.autograd.grad(y, (x, weight, bias), gy)
    elif case == 'gru':
        for c in (64, 32, 4):
            rows = lambda: guarded(torch.randn(2, 6, 5, c, generator=g), where).permute(0, 3, 1, 2).requires_grad_()
            pre_r, pre_u, h, cand = rows(), rows(), rows(), rows()
            bias = G(c)
            rh = tg.HipGruReset.apply(pre_r, bias, h, lib)
            hn = tg.HipGruOut.apply(pre_u, bias, h, cand, lib)
            gy = guarded(torch.randn(2, 6, 5, c, generator=g), where).permute(0, 3, 1, 2)
            torch.autograd.grad((rh, hn), (pre_r, pre_u, h, cand), (gy, gy))
    elif case == 'resample':
        for shape in ((2, 5, 7, 8), (1, 1, 1, 4), (3, 4, 1, 12), (2, 25, 25, 64)):
            x = guarded(torch.randn(*shape, generator=g), where).permute(0, 3, 1, 2).requires_grad_()
            y = tg.HipUpsample2x.apply(x, lib)
            gy = guarded(torch.randn(shape[0], 2 * shape[1], 2 * shape[2], shape[3], generator=g), where).permute(0, 3, 1, 2)
            torch.autograd.grad(y, x, gy)
            m = tg.HipSpatialMean.apply(x, lib)
            torch.autograd.grad(m, x, torch.randn(m.shape, generator=g))
    elif case == 'inputs':                               # input pipeline: instance labels, image preparation
        from fiery_amd.images import resize_crop_normalise
        from fiery_amd.labels import convert_instance_mask_to_center_and_offset_label
        ids = guarded(torch.randint(0, 6, (4, 20, 28), generator=g), where)
        ego = G(4, 6) * 0.1
        convert_instance_mask_to_center_and_offset_label(ids, ego, 5, spatial_extent=(10.0, 14.0), lib=lib, device='cpu')
        img = guarded(torch.randint(0, 256, (2, 45, 80, 3), generator=g, dtype=torch.uint8), where)
        resize_crop_normalise(img, (24, 13), (0, 2, 24, 14), lib=lib, device='cpu')
        resize_crop_normalise(img, (56, 31), (3, 0, 53, 31), lib=lib, device='cpu')
    elif case in ('pool', 'pool_compact_off'):           # voxel pooling forward (whatever form the library picks) and backward
        import numpy as np
        from oracle import lift_splat as ls
        from tests.test_kernels_sim_liftsplat import _grid, _small_problem
        if case == 'pool_compact_off':
            os.environ['FIERY_POOL_COMPACT'] = '0'
        for H, W, C, n_cam, big in ((28, 40, 2, 3, False), (27, 40, 4, 2, False), (6, 10, 3, 2, False), (28, 40, 2, 2, True)):
            frustum, intr, extr, lifted = _small_problem(60 + H, n_cam=n_cam, D=8, H=H, W=W, C=C, frames=2)
            frames, n_cam, C, D, H, W = lifted.shape
            geo = torch.from_numpy(ls.get_geometry(frustum, intr.numpy(), extr.numpy()))
            grid, _ = _grid([-14.0, 30.0, 0.125 if big else 0.5], [-24.0, 10.0, 0.125 if big else 0.5], [-10.0, 10.0, 20.0])
            x = guarded(lifted, where)
            st = x.stride()
            out = lib.voxel_pool(x, (st[0], st[1], st[3], st[4], st[5], st[2]), guarded(geo, where), frames, n_cam, D, H, W, C, grid)
            ws = lib.pool_workspace(frames, n_cam, D, H, W, x.device, grid)
            lib.voxel_pool(x, (st[0], st[1], st[3], st[4], st[5], st[2]), guarded(geo, where), frames, n_cam, D, H, W, C, grid, workspace=ws)
            gx = torch.empty(frames, n_cam, C, D, H, W).permute(0, 1, 3, 4, 5, 2)
            lib.voxel_pool_bwd(guarded(torch.randn(out.shape, generator=g), where), guarded(ws[:frames * n_cam * D * H * W], where), frames, n_cam,
                               D, H, W, C, gx)
            # the fused form: depth distribution and features apart
            prob = guarded(torch.rand(frames * n_cam, D, H, W, generator=g), where)
            feats = guarded(torch.randn(frames, n_cam, C, H, W, generator=g), where)
            lib.lift_splat(prob, feats, guarded(geo, where), frames, n_cam, D, H, W, C, grid)
    elif case == 'train_images':                         # a whole training step from images on a tiny configuration
        from fiery_amd.model import Fiery
        from fiery_amd.synthetic import make_inputs, randomise_weights
        from tests.helpers import tiny_cfg
        cfg = tiny_cfg('baseline.yml', bev=8, **{'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1, 'N_FUTURE_FRAMES': 1,
                                                  'TIME_RECEPTIVE_FIELD': 2, 'MODEL.ENCODER.NAME': 'efficientnet-b0'})
        torch.manual_seed(0)
        model = Fiery(cfg)
        randomise_weights(model)
        model.train()
        model._lib = lib
        # (one sample, two cameras: four images through the trunk - the simulator pays per launch and per work-item)
        image, K, E, ego = make_inputs(1, model.receptive_field + model.n_future, 2, image_hw=tuple(cfg.IMAGE.FINAL_DIM), seed=3)
        labels = torch.randn(1, 1 + model.n_future, 6, *model.bev_size, generator=g)
        out = model(image, K, E, ego, labels)
        sum((v.float() ** 2).mean() for v in out.values() if v is not None).backward()
    else:
        raise SystemExit(f'unknown case {case}')
    print('ok', case, where, f'({len(_keep)} guarded tensors)')


if __name__ == '__main__':
    main()
