"""Training path: the differentiable convolution on the HIP kernels and the training-mode
forward built from it, on the CPU-simulated kernels - against torch autograd and against the reference's own modules
run in train() mode."""
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import forward_case, randomise_weights, tiny_cfg


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


@pytest.mark.parametrize('cin,cout,k,stride,pad,hw', [(16, 24, 3, 1, 1, (9, 11)), (13, 8, 1, 1, 0, (6, 7)), (8, 40, 3, 2, 1, (9, 12)),
                                                      (8, 16, 3, 2, 1, (8, 10)), (11, 32, 7, 2, 3, (12, 12)), (70, 35, 3, 1, 1, (5, 6))])
def test_hip_conv2d_gradients_match_torch_autograd(sim, cin, cout, k, stride, pad, hw):
    from fiery_amd.train_graph import HipConv2d
    g = torch.Generator().manual_seed(cin * 31 + cout)
    x = torch.randn(2, cin, *hw, generator=g).requires_grad_()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_()
    y = HipConv2d.apply(x, w, stride, pad, sim)
    ref = F.conv2d(x, w, None, stride, pad)
    assert y.shape == ref.shape
    assert _rel(y, ref) < 2e-6
    gy = torch.randn(ref.shape, generator=g)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    rx, rw = torch.autograd.grad(ref, (x, w), gy)
    assert _rel(gx, rx) < 5e-6 and _rel(gw, rw) < 5e-6


def _grads_vs_fp64(fn_hip, fn_ref, inputs, seed, tol=2e-5):
    """Every output and every input gradient of an operator against the float64 autograd evaluation of its torch statement
    on well-conditioned inputs: relative L2 error <= tol PER TENSOR (a wrong bias gradient, a mis-scaled dgamma or a missing
    stride-2 tap is an O(1) error in one tensor - the 2 % median bound of the whole-step test would not see it)."""
    g = torch.Generator().manual_seed(seed)
    leaves32 = [t.clone().requires_grad_() for t in inputs]
    leaves64 = [t.double().clone().requires_grad_() for t in inputs]
    out32, out64 = fn_hip(*leaves32), fn_ref(*leaves64)
    assert out32.shape == out64.shape
    assert _rel(out32.double(), out64) < tol, ('output', _rel(out32.double(), out64))
    gy = torch.randn(out64.shape, generator=g, dtype=torch.float64)
    g32 = torch.autograd.grad(out32, leaves32, gy.float())
    g64 = torch.autograd.grad(out64, leaves64, gy)
    for i, (a, b) in enumerate(zip(g32, g64)):
        assert a.shape == b.shape
        assert _rel(a.double(), b) < tol, (f'gradient of input {i}', _rel(a.double(), b))


def test_gru_elementwise_operators_all_gradients_vs_fp64(sim):
    """HipGruReset / HipGruOut (layers/temporal.py:53-61): outputs and the gradients of the pre-activation, the BIAS, the state
    and the candidate."""
    from fiery_amd.train_graph import HipGruOut, HipGruReset
    g = torch.Generator().manual_seed(3)
    for c in (64, 32):                      # (the GRU's channel counts are multiples of 8: its kernels take 16-byte rows only)
        pre, h, cand = (torch.randn(2, c, 7, 9, generator=g) for _ in range(3))
        bias = 0.3 * torch.randn(c, generator=g)
        _grads_vs_fp64(lambda p, b, hh: HipGruReset.apply(p, b, hh, sim),
                       lambda p, b, hh: (1.0 - torch.sigmoid(p + b.view(1, -1, 1, 1))) * hh, [pre, bias, h], seed=c)
        _grads_vs_fp64(lambda p, b, hh, cc: HipGruOut.apply(p, b, hh, cc, sim),
                       lambda p, b, hh, cc: (1.0 - torch.sigmoid(p + b.view(1, -1, 1, 1))) * hh + torch.sigmoid(p + b.view(1, -1, 1, 1)) * cc,
                       [pre, bias, h, cand], seed=c + 1)


def test_upsampling_plane_mean_and_strided_convolution_gradients_vs_fp64(sim):
    from fiery_amd.train_graph import HipConv2d, HipSpatialMean, HipUpsample2x
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 12, 6, 7, generator=g)
    _grads_vs_fp64(lambda t: HipUpsample2x.apply(t, sim), lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False),
                   [x], seed=5)
    _grads_vs_fp64(lambda t: HipSpatialMean.apply(t, sim), lambda t: t.mean(dim=(2, 3)), [torch.randn(3, 35, 9, 11, generator=g)], seed=6)
    # stride 2 (zero-stuffed input gradient), odd and even maps, 3x3 and 7x7, padded channel counts
    for cin, cout, k, pad, hw in ((8, 16, 3, 1, (9, 12)), (16, 16, 3, 1, (13, 13)), (11, 32, 7, 3, (12, 10)), (24, 40, 1, 0, (8, 6))):
        xx = torch.randn(2, cin, *hw, generator=g)
        ww = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        _grads_vs_fp64(lambda a, b: HipConv2d.apply(a, b, 2, pad, sim), lambda a, b: F.conv2d(a, b, None, 2, pad), [xx, ww], seed=cin + k)


def test_bf16_operand_convolutions_in_the_training_graph(sim, monkeypatch):
    """`train_graph.CONV_PRECISION = bf16` (the library's mixed-precision mode for training): forward and input gradient
    are the convolutions of the bf16-ROUNDED operands with fp32 accumulation (exactly: compared at 3e-5 against torch on the
    rounded operands); the weight gradient of the 3 x 3 / stride 1 layers likewise, of the others in fp32."""
    from fiery_amd import native, train_graph as tg
    monkeypatch.setattr(tg, 'CONV_PRECISION', native.PRECISION_BF16)
    rb = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(12)
    for cin, cout, k, stride, pad, hw in ((32, 64, 3, 1, 1, (9, 14)), (64, 128, 3, 1, 1, (7, 5)), (64, 32, 1, 1, 0, (6, 7)), (32, 64, 3, 2, 1, (9, 12)),
                                          (96, 40, 3, 1, 1, (5, 120)), (16, 32, 3, 1, 1, (1, 9))):
        x = torch.randn(2, cin, *hw, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        xx, ww = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = tg.HipConv2d.apply(xx, ww, stride, pad, sim)
        gy = torch.randn(y.shape, generator=g)
        gx, gw = torch.autograd.grad(y, (xx, ww), gy)
        fwd_rounded = cin % 32 == 0                                # (whole 32-channel stages, as for the input gradient below)
        want_y = F.conv2d(rb(x) if fwd_rounded else x, rb(w) if fwd_rounded else w, None, stride, pad)
        assert torch.allclose(y, want_y, rtol=3e-5, atol=3e-5), (cin, cout, (y - want_y).abs().max())
        # (the input gradient is a convolution over the cout channels: the bf16 kernel takes it when those are whole
        # 32-channel stages, otherwise that launch runs the fp32 kernel - `fiery_conv_precision_used`)
        dgrad_rounded = cout % 32 == 0
        want_gx = torch.nn.grad.conv2d_input(x.shape, rb(w) if dgrad_rounded else w, rb(gy) if dgrad_rounded else gy, stride, pad)
        assert torch.allclose(gx, want_gx, rtol=3e-5, atol=3e-5), (cin, cout, (gx - want_gx).abs().max())
        # weight gradient: the 3 x 3 / stride 1 layers' kernel has a bf16 form (operands rounded, sixteen pixels per MFMA);
        # the other layers keep the fp32 kernels
        rounded = k == 3 and stride == 1
        want_gw = torch.nn.grad.conv2d_weight(rb(x) if rounded else x, w.shape, rb(gy) if rounded else gy, stride, pad)
        assert torch.allclose(gw, want_gw, rtol=1e-4, atol=1e-4), (cin, cout, (gw - want_gw).abs().max())
        if rounded:
            assert (gw - torch.nn.grad.conv2d_weight(x, w.shape, gy, stride, pad)).abs().max() > 1e-4
        assert not fwd_rounded or (y - F.conv2d(x, w, None, stride, pad)).abs().max() > 1e-4          # it IS the rounded-operand result


def test_two_source_convolution_and_channel_split_gradients_vs_fp64(sim):
    """HipConv2dCat (the GRU's convolutions over [x, h] without the concatenated copy: forward from two tensors, weight gradient
    per source, one input-gradient launch split in two views) and HipSplitChannels (update | reset halves of the fused gate
    convolution) against the float64 autograd evaluation of `conv2d(cat(...))` / `chunk`."""
    from fiery_amd.train_graph import HipConv2dCat, HipSplitChannels
    g = torch.Generator().manual_seed(17)
    for c0, c1, cout, k, hw in ((32, 64, 128, 3, (7, 9)), (64, 64, 64, 3, (6, 5)), (35, 24, 40, 1, (5, 8)), (8, 64, 128, 3, (9, 4))):
        x0, x1 = torch.randn(2, c0, *hw, generator=g), torch.randn(2, c1, *hw, generator=g)
        w = torch.randn(cout, c0 + c1, k, k, generator=g) / ((c0 + c1) * k * k) ** 0.5
        pad = (k - 1) // 2
        _grads_vs_fp64(lambda a, b, ww: HipConv2dCat.apply(a, b, ww, pad, sim), lambda a, b, ww: F.conv2d(torch.cat([a, b], 1), ww, None, 1, pad),
                       [x0, x1, w], seed=c0 + cout)
    x = torch.randn(2, 128, 6, 7, generator=g)
    w = torch.randn(3, generator=g)
    _grads_vs_fp64(lambda t, s: (lambda a, b: a * s[0] + b * s[1] * a)(*HipSplitChannels.apply(t)),
                   lambda t, s: (lambda a, b: a * s[0] + b * s[1] * a)(*t.chunk(2, dim=1)), [x, w], seed=3, tol=1e-6)


def test_maxpool_and_ego_warp_all_gradients_vs_fp64(sim):
    """HipMaxPool2x2 (the pooled skip of a down-sampling Bottleneck, layers/convolutions.py:150-166: odd sizes padded with a
    zero row / column that takes part in the maximum) and HipEgoWarp (`cumulative_warp_features`, utils/geometry.py:225-253:
    bilinear `grid_sample` with zero padding; the present frame is copied) against the float64 autograd evaluation of their
    torch statements - values and input gradients."""
    from fiery_amd.train_graph import HipEgoWarp, HipMaxPool2x2, cumulative_warp_features
    g = torch.Generator().manual_seed(8)
    for shape in ((2, 12, 6, 8), (1, 7, 7, 9), (2, 64, 13, 13), (1, 4, 1, 5)):
        x = torch.randn(*shape, generator=g)
        x[..., -1] = -1.0 - x[..., -1].abs()                    # odd widths: the zero padding wins its window
        x[:, :, 0, 0] = x[:, :, 0, 1]                           # a tie: the first element of the window takes the gradient
        _grads_vs_fp64(lambda t: HipMaxPool2x2.apply(t, sim),
                       lambda t: F.max_pool2d(F.pad(t, (0, t.shape[-1] % 2, 0, t.shape[-2] % 2), value=0), 2, 2), [x], seed=shape[1])
    for (b, s, c, h, w), extent in (((2, 3, 8, 12, 10), (6.0, 5.0)), ((1, 2, 64, 16, 16), (8.0, 8.0)), ((1, 3, 5, 7, 70), (3.5, 35.0))):
        x = torch.randn(b, s, c, h, w, generator=g)
        ego = torch.randn(b, s, 6, generator=g) * torch.tensor([1.5, 1.5, 0.1, 0.02, 0.02, 0.2])
        theta = sim.warp_params(ego.contiguous(), extent).reshape(b * s, 6)
        identity = [(i % s) == s - 1 for i in range(b * s)]
        _grads_vs_fp64(lambda t: HipEgoWarp.apply(t.reshape(b * s, c, h, w), theta, identity, sim).reshape(b, s, c, h, w),
                       lambda t: cumulative_warp_features(t.clone(), ego.double(), 'bilinear', extent), [x], seed=c, tol=5e-5)


@pytest.mark.parametrize('c,k,stride,pads,hw', [(8, 3, 1, (1, 1, 1, 1), (7, 9)), (12, 3, 2, (0, 0, 1, 1), (9, 12)), (8, 5, 1, (2, 2, 2, 2), (6, 7)),
                                                 (16, 5, 2, (1, 1, 2, 2), (11, 10)), (8, 5, 2, (2, 2, 2, 2), (8, 8)), (4, 3, 2, (0, 1, 1, 0), (5, 5)),
                                                 # stride 1, lopsided pads, widths that are not multiples of k (the register-window walk of the weight gradient)
                                                 (8, 3, 1, (0, 2, 2, 0), (5, 8)), (12, 5, 1, (1, 3, 3, 1), (7, 11)), (4, 5, 1, (4, 0, 0, 4), (3, 4))])
def test_depthwise_convolution_all_gradients_vs_fp64(sim, c, k, stride, pads, hw):
    """`HipDepthwiseConv2d` (the image trunk's MBConv depthwise layers, 'static same' padding - asymmetric): output, input
    gradient (stride 2: zero-stuffed gradient, mirrored taps; rows no window reaches get zeros) and weight gradient
    (`fiery_depthwise_conv_wgrad_nhwc`) against float64 autograd of pad + conv2d(groups = C)."""
    from fiery_amd.train_graph import HipDepthwiseConv2d
    g = torch.Generator().manual_seed(c * 7 + k + stride)
    x = torch.randn(2, c, *hw, generator=g)
    w = torch.randn(c, 1, k, k, generator=g) / k
    top, left, bottom, right = pads
    _grads_vs_fp64(lambda a, b: HipDepthwiseConv2d.apply(a, b, stride, pads, sim),
                   lambda a, b: F.conv2d(F.pad(a, (left, right, top, bottom)), b, None, stride, 0, 1, c), [x, w], seed=k)


def test_trunk_and_lift_head_on_the_training_graph_equal_the_torch_statement(sim):
    """`TrainGraph.lift_head` - EfficientNet-b4 stem + blocks 0-21 (expansion, depthwise, squeeze-and-excite, projection,
    drop-connect, skips), the x2 upsampling, the head's convolutions - on the HIP operators against `Encoder.lift_head`, the
    torch statement of the same layers, both in train() mode from the same generator state: outputs, running statistics and
    the gradient of every parameter of the encoder."""
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    from fiery_amd.train_graph import TrainGraph
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg)
    randomise_weights(model)
    model._lib = sim
    ref = Fiery(cfg)
    ref.load_state_dict(model.state_dict())
    model.train()
    ref.train()
    images = torch.randn(2, 3, 32, 64, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(5)
    d_ref, f_ref = ref.encoder.lift_head(images)
    torch.manual_seed(5)
    d_got, f_got = TrainGraph(model, sim).lift_head(images)
    assert _rel(d_got, d_ref) < 2e-4 and _rel(f_got, f_ref) < 2e-4, (_rel(d_got, d_ref), _rel(f_got, f_ref))
    gen = torch.Generator().manual_seed(2)
    gd, gf = torch.randn(d_ref.shape, generator=gen), torch.randn(f_ref.shape, generator=gen)
    (d_ref * gd).sum().add((f_ref * gf).sum()).backward()
    (d_got * gd).sum().add((f_got * gf).sum()).backward()
    worst = 0.0
    for (name, p), (_, q) in zip(model.encoder.named_parameters(), ref.encoder.named_parameters()):
        assert p.grad is not None, name
        # (the bias of a block's last BatchNorm feeds, through the skip chain, straight into the next batch-statistics
        # BatchNorm: its true gradient is zero and both evaluations hold ~1e-4 of rounding noise there - hence the absolute term)
        diff, scale = (p.grad - q.grad).norm().item(), q.grad.norm().item()
        err = diff / max(scale, 1e-12)
        worst = max(worst, min(err, diff / 1e-3))
        assert diff <= 5e-3 * scale + 1e-3, (name, err, scale)
    for (name, b), (_, c) in zip(model.encoder.named_buffers(), ref.encoder.named_buffers()):
        assert torch.allclose(b.float(), c.float(), rtol=1e-4, atol=1e-5), name
    assert worst > 0.0


def test_batchnorm_act_all_gradients_vs_fp64(sim):
    """HipBatchNormAct in training mode with and without ReLU: y, dx, dgamma, dbeta (inputs kept away from the ReLU's kink: a
    gate within rounding of zero is the one thing fp32 and fp64 may legitimately disagree on)."""
    from fiery_amd.train_graph import HipBatchNormAct
    g = torch.Generator().manual_seed(7)
    for c, relu in ((64, False), (35, True), (6, True)):
        x = torch.randn(3, c, 8, 9, generator=g)
        gamma, beta = 0.5 + torch.rand(c, generator=g), 0.2 * torch.randn(c, generator=g)

        def ref(t, ga, be, relu=relu):
            y = F.batch_norm(t, None, None, ga, be, True, 0.0, 1e-5)
            return F.relu(y) if relu else y
        if relu:                                     # push pre-activations at least 1e-3 away from zero
            with torch.no_grad():
                y0 = ref(x.double(), gamma.double(), beta.double(), relu=False)
                beta_shift = torch.where(y0.abs().amin(dim=(0, 2, 3)) < 1e-3, torch.full_like(beta, 2e-3).double(), torch.zeros_like(beta).double())
            beta = (beta.double() + beta_shift).float()
        _grads_vs_fp64(lambda t, ga, be: HipBatchNormAct.apply(t, ga, be, None, None, True, 0.0, 1e-5, relu, sim), ref, [x, gamma, beta],
                       seed=c, tol=5e-5)


@pytest.mark.parametrize('kt', [1, 2])
def test_causal_conv3d_as_time_shifted_2d_convolutions(sim, kt):
    from fiery_amd.train_graph import TrainGraph
    g = torch.Generator().manual_seed(kt)
    x = torch.randn(2, 16, 3, 6, 7, generator=g).requires_grad_()
    w = (torch.randn(8, 16, kt, 3, 3, generator=g) * 0.1).requires_grad_()
    tg = TrainGraph(None, sim)
    y = tg.conv3d_frames(x, w)
    ref = F.conv3d(F.pad(x, (1, 1, 1, 1, kt - 1, 0)), w)
    assert _rel(y, ref) < 2e-6
    gy = torch.randn(ref.shape, generator=g)
    for a, b in zip(torch.autograd.grad(y, (x, w), gy), torch.autograd.grad(ref, (x, w), gy)):
        assert _rel(a, b) < 5e-6


def _train_cfg(preset='baseline.yml', bev=16, **extra):
    over = {'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1, 'N_FUTURE_FRAMES': 1}
    over.update(extra)
    return tiny_cfg(preset, bev=bev, **over)


def _loss(out):
    """A fixed scalar of every output (seeded weights), so each head and the distribution parameters get gradient."""
    g = torch.Generator().manual_seed(123)
    total = 0.0
    for k in sorted(out):
        v = out[k]
        if v is None:
            continue
        total = total + (v * torch.randn(v.shape, generator=g)).sum()
    return total


def _torch_conv(x, w, stride, pad, lib):
    """The operator `HipConv2d` implements, stated with torch (checked against it tightly above)."""
    return F.conv2d(x, w, None, stride, pad)


def _reference_step(ref, cfg, state, lifted, K, E, ego, labels, noise):
    """One training-mode forward + backward of the reference from the lifted features."""
    theirs = ref.Fiery(cfg)
    theirs.load_state_dict(state)
    theirs.train()
    leaf = lifted.clone().requires_grad_()
    B, S, n = lifted.shape[:3]
    flat = leaf.reshape(B * S, n, *lifted.shape[3:]).permute(0, 1, 3, 4, 5, 2)
    theirs.encoder_forward = lambda x: flat                      # instance attribute; reference files untouched
    out = theirs(torch.zeros(B, K.shape[1], n, 3, 2, 2), K, E, ego, labels, noise)
    _loss(out).backward()
    return theirs, out, leaf.grad


def graph_step(cfg, state, lib, conv, dtype, lifted, K, E, ego, labels, noise, device='cpu', pool_device=None, plane_means=True):
    """One training-mode forward + backward of `fiery_amd.train_graph`: pooling on the kernels (fp32, on `pool_device`), the
    rest of the graph on `device` in `dtype` with the convolution `conv` (None: `HipConv2d`)
    -> (model, outputs, d loss / d lifted, parameter gradients), results as fp64 on the host."""
    from fiery_amd.model import Fiery
    from fiery_amd.train_graph import TrainGraph
    pool_device = pool_device or device
    model = Fiery(cfg)
    model.load_state_dict(state)
    model = model.train().to(device=device, dtype=dtype)
    model._lib = lib
    pooler = model
    if str(pool_device) != str(device) or dtype != torch.float32:
        pooler = Fiery(cfg).to(pool_device)                                  # geometry + pooling only: no weights involved
        pooler._lib = lib
    rf = model.receptive_field
    leaf = lifted.clone().to(pool_device).requires_grad_()
    bev = TrainGraph(pooler, lib)._pooled(K[:, :rf].contiguous().to(pool_device), E[:, :rf].contiguous().to(pool_device), leaf[:, :rf])
    cast = lambda t: None if t is None else t.to(device=device, dtype=dtype)
    graph = TrainGraph(model, lib, conv2d=conv)
    graph.whole_plane_pooling_as_means = plane_means
    out = graph.bev_stack(cast(bev), cast(ego[:, :rf]), cast(labels), cast(noise))
    g = torch.Generator().manual_seed(123)
    total = 0.0
    for k in sorted(out):
        if out[k] is not None:
            total = total + (out[k] * torch.randn(out[k].shape, generator=g).to(device=device, dtype=dtype)).sum()
    total.backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return model, {k: (None if v is None else v.detach().double().cpu()) for k, v in out.items()}, leaf.grad.double().cpu(), grads


def as_close_to_exact_as_fp32_torch(hip, torch32, exact, test='train_step'):
    """BatchNorm over the few values per channel a small configuration leaves makes the training step ill-conditioned, and
    a ReLU / max-pool whose argument sits within rounding of a tie gates differently in two fp32 evaluations (one flipped
    gate moves every gradient upstream of it by a fraction of a percent).  A fixed tolerance against an fp32 reference would
    measure that reference's rounding, so the yardstick is the fp64 evaluation of the same graph: per tensor, the HIP path's
    relative L2 distance to it is at most 3x the all-torch fp32 evaluation's (+1e-4) or 2 % - for all but a tenth of the
    tensors, which may sit upstream of a flipped gate (none beyond 50 %: a kernel or layout mistake is an O(1) error in
    everything it touches); median and worst go to the parity ledger."""
    from tests import parity_report
    _, out_h, dx_h, g_h = hip
    _, out_t, dx_t, g_t = torch32
    _, out_e, dx_e, g_e = exact
    for k, v in out_e.items():
        if v is not None:
            err, yard, scale = (out_h[k] - v).abs().max().item(), (out_t[k] - v).abs().max().item(), max(1.0, v.abs().max().item())
            parity_report.record(test, k, (out_h[k] - out_t[k]).abs().max().item(), scale, err, yard, 3 * yard + 1e-4 * scale)
            assert err <= 3 * yard + 1e-4 * scale, k
    top = max(v.norm().item() / v.numel() ** 0.5 for v in g_e.values())
    assert set(g_h) == set(g_e)
    rows = []
    for name, a, b, e in [('d lifted', dx_h, dx_t, dx_e)] + [(n, g_h[n], g_t[n], g_e[n]) for n in g_e]:
        scale = e.norm().item() + 1e-4 * top * e.numel() ** 0.5      # floor: gradients that are zero in exact arithmetic
        rows.append(((a - e).norm().item() / scale, (b - e).norm().item() / scale, name))
    errs = sorted(r[0] for r in rows)
    median, worst = errs[len(errs) // 2], max(rows)
    flipped = [r for r in rows if r[0] > max(3 * r[1] + 1e-4, 2e-2)]
    parity_report.record(test, f'gradients: {len(rows)} tensors, relative L2, median', median, 1.0, median, sorted(r[1] for r in rows)[len(rows) // 2],
                         2e-2, 'against the fp64 graph; yardstick = all-torch fp32 graph')
    parity_report.record(test, f'gradients: worst ({worst[2][-48:]}); {len(flipped)} past 2 %', worst[0], 1.0, worst[0], worst[1], 0.5)
    # (the yardstick applies to the median too: on the tiny simulator configuration the all-torch fp32 graph's own median is
    # 2.8 % - a 1e-7 change of a sampling position moves these gradients by percents; a fixed 2 % would measure the seed)
    median_torch = sorted(r[1] for r in rows)[len(rows) // 2]
    assert median <= max(2e-2, 1.25 * median_torch) and worst[0] <= 0.5 and len(flipped) <= len(rows) // 10, (median, median_torch, worst, len(flipped))
    # THE criterion, per tensor and without allowances: no gradient tensor of the kernels' step is further from the fp64
    # evaluation than a small multiple of what the all-torch fp32 evaluation of the same graph is (+ 1e-3 of the tensor's norm).  A kernel,
    # layout or indexing mistake is an O(1) error in every tensor it touches and cannot hide behind that; what the bound
    # does absorb is exactly what fp32 arithmetic itself does to this step (on the tiny configuration PyTorch's own fp32
    # graph is 20 % from fp64 on some tensors and the kernels' 6 %).  The looser clauses above stay as a second net.
    # (One documented exception: the parameters of the pyramid pooling's 1 x 1 x 1 convolution sit directly in front of a
    # train-mode BatchNorm over B x T pooled values - three at B = 1 - whose backward subtracts nearly equal numbers; there two
    # fp32 evaluations differ by percents whatever computes them (MI355X, 104 x 104, B = 1: 1.7e-2 here, 5e-4 for ATen's
    # order of operations, 0 violations at B = 2).  They get 5 %.)
    # (The factor is THREE, and the yardstick deterministic.  Round 4 widened it to ten after one GPU run in five failed "by a
    # tenth": the kernels' own columns of the ledger were the same to every digit in five runs (median 7.559e-3, worst
    # 1.655e-2), but the all-torch fp32 graph evaluated on the GPU moved by 10-20 % from run to run (ATen's GPU reductions) and
    # a tensor's bound with it.  Since round 5 the GPU test evaluates the yardstick graph on the host with a fixed thread count
    # - same numbers every run - so the bound no longer needs room for a draw of the yardstick.)
    bound = lambda r, factor: max(factor * r[1] + 1e-3, 5e-2 if 'pyramid_pooling' in r[2] else 0.0)
    past3 = sorted((r for r in rows if r[0] > bound(r, 3)), reverse=True)
    strict = past3
    parity_report.record(test, f'gradients: {len(rows) - len(past3)} of {len(rows)} tensors within 3x the torch fp32 graph\'s distance to fp64 (+1e-3); asserted: all',
                         past3[0][0] if past3 else 0.0, 1.0, past3[0][0] if past3 else 0.0, past3[0][1] if past3 else 0.0, 0.0,
                         '; '.join(f'{r[2][-40:]} {r[0]:.4f} (torch {r[1]:.4f})' for r in past3[:6]))
    if os.environ.get('FIERY_TEST_VERBOSE'):
        for r in sorted(rows, reverse=True)[:40]:
            print(f'{r[2]:80s} hip {r[0]:9.2e} torch {r[1]:9.2e}')
    assert not strict, strict[:8]
    # The 1 % question, per tensor: where fp32 arithmetic itself allows it - the all-torch fp32 graph is within 0.5 % of the
    # fp64 one - the kernels' gradient is within 1 % (a flipped gate may again take a tenth of the tensors out); the tensors
    # past 1 % are listed with the torch figure beside them, which is what says whether conditioning or a kernel is the cause.
    past = sorted((r for r in rows if r[0] > 1e-2), reverse=True)
    conditioned = [r for r in rows if r[1] <= 5e-3]
    unexplained = [r for r in conditioned if r[0] > 1e-2]
    parity_report.record(test, f'gradients: {len(rows) - len(past)} of {len(rows)} tensors within 1 % of fp64; of the {len(past)} past it '
                         f'{sum(1 for r in past if r[1] > 5e-3)} have the torch fp32 graph past 0.5 % too', len(past) / len(rows), 1.0,
                         past[0][0] if past else 0.0, past[0][1] if past else 0.0, 1e-2,
                         'past 1 %: ' + '; '.join(f'{r[2][-40:]} {r[0]:.3f} (torch {r[1]:.3f})' for r in past[:6]))
    assert len(unexplained) <= len(rows) // 10, unexplained[:8]
    return len(g_e)


def _case(preset, B=2, bev=16, **extra):
    from fiery_amd.model import Fiery
    cfg = _train_cfg(preset, bev=bev, **extra)
    torch.manual_seed(0)
    model = Fiery(cfg)
    state = {k: v.clone() for k, v in randomise_weights(model).items()}
    inputs = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, B, 2,
                          with_labels=model.n_future > 0, with_noise=model.n_future > 0)
    return cfg, state, inputs


@pytest.mark.needs_reference
@pytest.mark.parametrize('preset', ['baseline.yml', 'literature/static_lss_setting.yml', 'temporal_single_timeframe.yml'])
def test_training_graph_is_the_reference_in_train_mode(sim, preset):
    """Wiring: `train_graph` (with torch's convolution standing in for the kernel and the pyramid pooling taken operator for
    operator, so both sides round alike; the means form of that pooling is checked against this one below) against the
    reference's modules in train() mode - batch-statistics BatchNorm, latent sampled from the FUTURE distribution: the
    outputs, the gradient of every BEV-stack parameter and of the lifted features, the running statistics after the step."""
    from oracle.ref_shims import load_reference
    ref = load_reference()
    cfg, state, inputs = _case(preset, bev=48)          # (both sides are torch operators on the host: a healthier size is cheap)
    theirs, want, want_dx = _reference_step(ref, cfg, state, *inputs)
    ours, got, got_dx, grads = graph_step(cfg, state, sim, _torch_conv, torch.float32, *inputs, plane_means=False)
    for k, v in want.items():
        if v is None:
            assert got[k] is None
        else:
            assert _rel(got[k], v.double()) < 2e-4, k        # fp32 BatchNorm statistics of 14k values summed in another order
    theirs_grads = {n: p.grad.double() for n, p in theirs.named_parameters() if p.grad is not None}
    assert set(theirs_grads) == set(grads)
    top = max(v.norm().item() / v.numel() ** 0.5 for v in theirs_grads.values())
    rel_l2 = lambda a, e: (a - e).norm().item() / (e.norm().item() + 1e-4 * top * e.numel() ** 0.5)    # floor: exactly-zero gradients
    # (with torch's operators on both sides autograd derives the backward pass from a forward pass that is already shown
    # equal; what is left is rounding - a ReLU / max-pool gate of a 3 x 3 map flipping between the two fp32 evaluations
    # moves everything upstream by a percent or two)
    assert rel_l2(got_dx, want_dx.double()) < 5e-2
    for name, g in theirs_grads.items():
        assert rel_l2(grads[name], g) < 5e-2, name
    assert len(grads) > (50 if ours.n_future > 0 else 20)
    theirs_buffers = dict(theirs.named_buffers())
    for name, b in ours.named_buffers():
        if name.endswith(('running_mean', 'running_var', 'num_batches_tracked')) and not name.startswith('encoder.'):
            assert torch.allclose(b.float(), theirs_buffers[name].float(), rtol=1e-4, atol=1e-5), name


def test_whole_plane_pooling_as_means_is_the_pooling_operator(sim):
    """The (2, H, W) pyramid pooling as per-frame means + a broadcast against avg_pool3d + bilinear interpolation, both
    in fp64: the same function, values and gradients."""
    cfg, state, inputs = _case('baseline.yml')
    a = graph_step(cfg, state, sim, _torch_conv, torch.float64, *inputs, plane_means=True)
    b = graph_step(cfg, state, sim, _torch_conv, torch.float64, *inputs, plane_means=False)
    for k, v in b[1].items():
        if v is not None:
            assert torch.allclose(a[1][k], v, rtol=1e-9, atol=1e-9), k
    assert torch.allclose(a[2], b[2], rtol=1e-7, atol=1e-9 * b[2].abs().max().item())
    for name, g in b[3].items():
        assert torch.allclose(a[3][name], g, rtol=1e-7, atol=1e-8 * max(1.0, g.abs().max().item())), name


@pytest.mark.parametrize('preset', ['baseline.yml'])
def test_training_step_on_the_kernels_is_as_close_to_exact_as_fp32_torch(sim, preset):
    """Kernels: the same graph with `HipConv2d` (forward, input gradient and weight gradient on the simulated kernels)
    against its own fp64 evaluation, bounded by the error of the all-torch fp32 evaluation."""
    cfg, state, inputs = _case(preset, **{'TIME_RECEPTIVE_FIELD': 2})          # (one temporal block: the simulator pays per launch)
    exact = graph_step(cfg, state, sim, _torch_conv, torch.float64, *inputs)
    torch32 = graph_step(cfg, state, sim, _torch_conv, torch.float32, *inputs)
    hip = graph_step(cfg, state, sim, None, torch.float32, *inputs)
    assert as_close_to_exact_as_fp32_torch(hip, torch32, exact, 'train_step[sim tiny]') > 50
    # BatchNorm's side effects on the HIP operators: running statistics as the torch statement's, and every layer's batch counter
    # (collected over the pass and applied in one launch, train_graph._counts_batches_once) up by its number of calls - the GRU's
    # norm runs once per future frame
    want = dict(torch32[0].named_buffers())
    counters = 0
    for name, b in hip[0].named_buffers():
        if name.endswith('num_batches_tracked') and not name.startswith('encoder.'):
            assert int(b) == int(want[name]), name
            counters += int(b) > 0
        elif name.endswith(('running_mean', 'running_var')) and not name.startswith('encoder.'):
            assert torch.allclose(b, want[name], rtol=1e-3, atol=1e-5), name
    assert counters > 20 and max(int(b) for n, b in hip[0].named_buffers() if n.endswith('num_batches_tracked')) == hip[0].n_future


@pytest.mark.gpu
def test_zeroed_chunks_hand_out_disjoint_zero_slices(hip):
    """`Lib.zeros_f32`: the weight gradients' accumulate-into outputs as slices of chunks zeroed in one fill - zero, 256-byte
    aligned, disjoint, never handed out twice (a held view keeps its values when the chunk runs out), per stream; too large or
    under capture: a plain allocation."""
    dev = torch.device('cuda:0')
    a = hip.zeros_f32((64, 9, 64), dev)
    b = hip.zeros_f32((3, 5), dev)
    assert a.is_contiguous() and a.shape == (64, 9, 64) and a.data_ptr() % 256 == 0 and b.data_ptr() % 256 == 0
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0
    a.fill_(1.0)
    assert float(b.abs().sum()) == 0.0 and b.data_ptr() >= a.data_ptr() + a.numel() * 4
    held = [a, b]                                         # (held: a chunk whose views are all gone goes back to the allocator)
    for _ in range(3 * hip.ZERO_CHUNK_FLOATS // (64 * 9 * 64)):                 # through several chunks
        held.append(hip.zeros_f32((64, 9, 64), dev))
    assert len({v.data_ptr() for v in held}) == len(held)
    assert float(torch.stack(held[2:]).abs().sum()) == 0.0 and float(a.sum()) == a.numel() and float(b.abs().sum()) == 0.0
    big = hip.zeros_f32((hip.ZERO_CHUNK_FLOATS,), dev)
    assert float(big.abs().sum()) == 0.0 and big.numel() == hip.ZERO_CHUNK_FLOATS
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        c = hip.zeros_f32((8, 8), dev)
        c.add_(2.0)
    side.synchronize()
    d = hip.zeros_f32((8, 8), dev)
    assert float(c.sum()) == 128.0 and float(d.abs().sum()) == 0.0
    assert hip.zeros_f32((4, 4), torch.device('cpu')).device.type == 'cpu'


def test_frame_views_and_splits_hand_back_pixel_major_gradients():
    """`train_graph._frames_view` / `_Frames`: the same values as `view(b, t, ...)[:, k:]` / `x[:, t]`, but the gradient autograd
    materialises for them is laid out as pixel-major rows like the activations - an NCHW gradient there made every accumulation
    below it a transposing add and every BatchNorm backward start with a copy (round 5: 4.7 ms of a 55 ms step)."""
    from fiery_amd.train_graph import _Frames, _frames_view, _stack_frames
    g = torch.Generator().manual_seed(5)
    leaf = torch.randn(6, 8, 5, 7, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_()
    weight = torch.randn(2, 2, 8, 5, 7, generator=g)
    seen = {}
    for name, view in (('ours', lambda z: _frames_view(z, 2, 3, 1)), ('plain', lambda z: z.view(2, 3, 8, 5, 7)[:, 1:])):
        z = leaf * 1.0
        assert z.is_contiguous(memory_format=torch.channels_last)
        z.register_hook(lambda grad, name=name: seen.__setitem__(name, grad))
        v = view(z)
        assert v.shape == (2, 2, 8, 5, 7) and torch.equal(v, leaf.detach().view(2, 3, 8, 5, 7)[:, 1:])
        (v * weight).sum().backward()
    assert torch.equal(seen['ours'], seen['plain'])
    assert seen['ours'].is_contiguous(memory_format=torch.channels_last) and not seen['plain'].is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(_frames_view(leaf, 2, 3), leaf.view(2, 3, 8, 5, 7))
    # frame splits: one pixel-major stack of the frame gradients, zeros for a frame nobody used
    frames5 = _stack_frames([torch.randn(2, 8, 5, 7, generator=g) for _ in range(3)]).requires_grad_()
    x = frames5 * 1.0
    x.register_hook(lambda grad: seen.__setitem__('split', grad))
    parts = _Frames.apply(x)
    assert len(parts) == 3 and all(torch.equal(parts[t], frames5.detach()[:, t]) for t in range(3))
    (parts[0] * 2.0).sum().backward(retain_graph=True)
    want = torch.zeros(2, 3, 8, 5, 7)
    want[:, 0] = 2.0
    assert torch.equal(seen['split'], want) and seen['split'].permute(0, 1, 3, 4, 2).is_contiguous()
    (parts[0] * 2.0 + parts[2] * parts[2]).sum().backward()
    want[:, 2] = 2.0 * frames5.detach()[:, 2]
    assert torch.allclose(seen['split'], want)


def test_training_mode_needs_the_future_labels(sim):
    from fiery_amd.model import Fiery
    cfg = _train_cfg()
    torch.manual_seed(0)
    model = Fiery(cfg).train()
    model._lib = sim
    lifted, K, E, ego, labels, noise = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, 2, 2)
    with pytest.raises(ValueError, match='future distribution'):
        model.bev_forward(lifted.requires_grad_(), K, E, ego, None, None)


def test_forward_from_images_in_training_mode_reaches_every_parameter(sim):
    """`Fiery.forward` under model.train(): image trunk + lift head as torch modules, fused lift-splat (HIP forward and
    backward), the training graph - one optimiser step changes the weights and the next pass still runs (nothing cached
    from the old weights)."""
    from fiery_amd.model import Fiery
    from fiery_amd.synthetic import make_inputs
    # (EfficientNet-b0, the reference's other trunk (encoder.py:40-56): a third of the b4 trunk's work on the simulator; the b4
    # trunk under autograd is covered by test_trunk_and_lift_head_on_the_training_graph_equal_the_torch_statement)
    cfg = tiny_cfg('baseline.yml', bev=8, **{'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1,
                                             'N_FUTURE_FRAMES': 1, 'TIME_RECEPTIVE_FIELD': 2, 'MODEL.ENCODER.NAME': 'efficientnet-b0'})
    torch.manual_seed(0)
    model = Fiery(cfg)
    randomise_weights(model)
    model.train()
    model._lib = sim
    B, n = 1, 2                                  # (two cameras of one sample: half the images of a batch of two, both loops exercised)
    image, K, E, ego = make_inputs(B, model.receptive_field + model.n_future, n, image_hw=tuple(cfg.IMAGE.FINAL_DIM), seed=3)
    labels = torch.randn(B, 1 + model.n_future, 6, *model.bev_size, generator=torch.Generator().manual_seed(4))
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    out = model(image, K, E, ego, labels)
    loss = sum((v.float() ** 2).mean() for v in out.values() if v is not None)
    loss.backward()
    missing = [name for name, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt.step()
    assert any(not torch.equal(p, before[n]) for n, p in model.named_parameters())


# ---- on the MI355X ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,k,stride,pad,hw', [(64, 64, 3, 1, 1, (200, 200)), (70, 35, 3, 1, 1, (100, 100)), (64, 64, 7, 2, 3, (200, 200)),
                                                      (128, 128, 3, 2, 1, (25, 25)), (32, 2, 1, 1, 0, (200, 200)), (256, 128, 1, 1, 0, (50, 50))])
def test_hip_conv2d_real_shapes_against_fp64(hip, cin, cout, k, stride, pad, hw):
    """Forward, input gradient and weight gradient of `HipConv2d` at the path's own layer shapes against an fp64 evaluation
    on the host (the fp32 operator of PyTorch-ROCm beside it as the yardstick)."""
    from fiery_amd.train_graph import HipConv2d
    from tests import parity_report
    g = torch.Generator().manual_seed(cin * 31 + cout)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    gy = None
    results = {}
    for name, dev, dt, fn in (('exact', 'cpu', torch.float64, None), ('torch32', 'cuda', torch.float32, None),
                              ('hip', 'cuda', torch.float32, HipConv2d.apply)):
        xx, ww = x.to(device=dev, dtype=dt).requires_grad_(), w.to(device=dev, dtype=dt).requires_grad_()
        y = fn(xx, ww, stride, pad, hip) if fn else F.conv2d(xx, ww, None, stride, pad)
        if gy is None:
            gy = torch.randn(y.shape, generator=g)
        gx, gw = torch.autograd.grad(y, (xx, ww), gy.to(device=dev, dtype=dt))
        results[name] = [t.detach().double().cpu() for t in (y, gx, gw)]
    for i, what in enumerate(('y', 'dx', 'dw')):
        exact = results['exact'][i]
        err_hip = (results['hip'][i] - exact).abs().max().item()
        err_t = (results['torch32'][i] - exact).abs().max().item()
        scale = exact.abs().max().item()
        parity_report.record(f'hip_conv2d[{cin}>{cout} k{k} s{stride} {hw[0]}]', what,
                             (results['hip'][i] - results['torch32'][i]).abs().max().item(), scale, err_hip, err_t,
                             3 * err_t + 1e-5 * scale, 'reference = the fp32 operator of PyTorch-ROCm; bound is on the fp64 error')
        assert err_hip <= 3 * err_t + 1e-5 * scale, (what, err_hip, err_t, scale)


@pytest.mark.gpu
def test_maxpool_and_ego_warp_real_shapes_against_fp64(hip):
    """`HipMaxPool2x2` at the distribution encoder's sizes (200 -> 100 -> 50 -> 25 -> 13: the last one odd) and `HipEgoWarp` at
    baseline.yml's (2 samples x 3 frames x 64 channels x 200 x 200): values and input gradients against the float64 evaluation
    of the torch statement on the host, with PyTorch-ROCm's fp32 operators on the same GPU as the yardstick."""
    from fiery_amd.train_graph import HipEgoWarp, HipMaxPool2x2, cumulative_warp_features
    from tests import parity_report
    g = torch.Generator().manual_seed(21)

    def check(name, run, x, gy=None):
        results = {}
        for kind, dev, dt in (('exact', 'cpu', torch.float64), ('torch32', 'cuda', torch.float32), ('hip', 'cuda', torch.float32)):
            xx = x.to(device=dev, dtype=dt).requires_grad_()
            y = run(xx, kind == 'hip')
            if gy is None:
                gy = torch.randn(y.shape, generator=g)
            gx, = torch.autograd.grad(y, xx, gy.to(device=dev, dtype=dt))
            results[kind] = [t.detach().double().cpu() for t in (y, gx)]
        for i, what in enumerate(('y', 'dx')):
            exact = results['exact'][i]
            err_hip, err_t = (results['hip'][i] - exact).abs().max().item(), (results['torch32'][i] - exact).abs().max().item()
            scale = exact.abs().max().item()
            parity_report.record(name, what, (results['hip'][i] - results['torch32'][i]).abs().max().item(), scale, err_hip, err_t,
                                 3 * err_t + 1e-5 * scale, 'reference = the fp32 operator of PyTorch-ROCm; bound is on the fp64 error')
            assert err_hip <= 3 * err_t + 1e-5 * scale, (name, what, err_hip, err_t, scale)

    for c, hw in ((70, (200, 200)), (64, (25, 25)), (128, (13, 13))):
        x = torch.randn(2, c, *hw, generator=g)
        check(f'hip_maxpool[{c} {hw[0]}]',
              lambda t, ours: HipMaxPool2x2.apply(t, hip) if ours else F.max_pool2d(F.pad(t, (0, t.shape[-1] % 2, 0, t.shape[-2] % 2), value=0), 2, 2), x)
    b, s, c, h, w, extent = 2, 3, 64, 200, 200, (50.0, 50.0)
    x = torch.randn(b, s, c, h, w, generator=g)
    ego = torch.randn(b, s, 6, generator=g) * torch.tensor([2.0, 2.0, 0.1, 0.01, 0.01, 0.1])
    theta = hip.warp_params(ego.cuda().contiguous(), extent).reshape(b * s, 6)
    identity = [(i % s) == s - 1 for i in range(b * s)]
    check('hip_ego_warp[2x3x64x200x200]',
          lambda t, ours: (HipEgoWarp.apply(t.reshape(b * s, c, h, w), theta, identity, hip).reshape(b, s, c, h, w) if ours else
                           cumulative_warp_features(t.clone(), ego.to(device=t.device, dtype=t.dtype), 'bilinear', extent)), x)


@pytest.mark.gpu
@pytest.mark.parametrize('c,k,stride,hw', [(144, 3, 2, (112, 240)), (192, 5, 2, (56, 120)), (336, 3, 1, (28, 60)), (960, 5, 1, (14, 30))])
def test_depthwise_convolution_real_shapes_against_fp64(hip, c, k, stride, hw):
    """`HipDepthwiseConv2d` at the image trunk's own layer shapes (EfficientNet-b4 at 224 x 480, six images) with the trunk's
    'static same' padding: output, input gradient and weight gradient against float64 autograd on the host."""
    from fiery_amd.train_graph import HipDepthwiseConv2d
    from tests import parity_report
    g = torch.Generator().manual_seed(c + k)
    h, w = hw
    pad_h = max((-(-h // stride) - 1) * stride + k - h, 0)
    pad_w = max((-(-w // stride) - 1) * stride + k - w, 0)
    pads = (pad_h // 2, pad_w // 2, pad_h - pad_h // 2, pad_w - pad_w // 2)            # top, left, bottom, right
    x = torch.randn(6, c, h, w, generator=g)
    wt = torch.randn(c, 1, k, k, generator=g) / k
    xe, we = x.double().requires_grad_(), wt.double().requires_grad_()
    ye = F.conv2d(F.pad(xe, (pads[1], pads[3], pads[0], pads[2])), we, None, stride, 0, 1, c)
    gy = torch.randn(ye.shape, generator=g)
    gxe, gwe = torch.autograd.grad(ye, (xe, we), gy.double())
    xh, wh = x.cuda().requires_grad_(), wt.cuda().requires_grad_()
    yh = HipDepthwiseConv2d.apply(xh, wh, stride, pads, hip)
    gxh, gwh = torch.autograd.grad(yh, (xh, wh), gy.cuda())
    for what, got, want in (('y', yh, ye), ('dx', gxh, gxe), ('dw', gwh, gwe)):
        err = (got.detach().double().cpu() - want.detach()).abs().max().item()
        scale = want.abs().max().item()
        parity_report.record(f'hip_depthwise[{c} k{k} s{stride} {h}x{w}]', what, err, scale, err, None, 2e-5 * scale)
        assert err <= 2e-5 * scale, (what, err, scale)


@pytest.mark.gpu
def test_training_step_on_the_gpu_is_as_close_to_exact_as_fp32_torch(hip):
    """The kernels test of the CPU tier on the real kernels at a quarter-size BEV grid (104 x 104 - the decoder's x2 stages need a multiple of 8 -, full image size, six
    cameras): the HIP training step against the fp64 evaluation of the same graph (PyTorch-ROCm operators in double), bounded by the all-torch fp32
    evaluation's error."""
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    cfg = get_preset_cfg('baseline.yml', ['LIFT.X_BOUND', '[-26.0, 26.0, 0.5]', 'LIFT.Y_BOUND', '[-26.0, 26.0, 0.5]', 'N_FUTURE_FRAMES', '2'])
    torch.manual_seed(0)
    model = Fiery(cfg)
    state = {k: v.clone() for k, v in randomise_weights(model).items()}
    inputs = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, 1, 6, with_labels=True,
                          with_noise=True)
    exact = graph_step(cfg, state, hip, _torch_conv, torch.float64, *inputs, device='cuda')
    # The yardstick - the all-torch fp32 evaluation of the same graph - runs on the HOST with a fixed thread count (pooling on the
    # kernels, like the other two): ATen's CPU reductions are deterministic for a given thread count, its GPU reductions are not
    # (round 4: the same tensor's torch distance moved by 10-20 % from run to run, and with it a bound of 3x that distance).
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        torch32 = graph_step(cfg, state, hip, _torch_conv, torch.float32, *inputs, device='cpu', pool_device='cuda')
    finally:
        torch.set_num_threads(threads)
    got = graph_step(cfg, state, hip, None, torch.float32, *inputs, device='cuda')
    assert as_close_to_exact_as_fp32_torch(got, torch32, exact, 'train_step[baseline 104x104 B1]') > 100


@pytest.mark.gpu
def test_full_size_training_steps_reduce_the_loss(hip):
    """In-process (round 2 ran it in a child process: its backward pass died with "Memory access fault by GPU" after the
    ~70 other GPU tests).  Root cause, found in round 3 with GPU guard pages (tools/guard_alloc): MIOpen's
    `igemm_bwd_gtcx35_nhwc_fp32_*` - the backward-data kernel PyTorch-ROCm's convolutions of the image trunk get - reads past
    the end of its operand; whether that touches an unmapped page depends on where the caching allocator put the tensor.
    `fiery_amd/__init__.py` excludes that solver; DESIGN.md section 9c has the evidence."""
    _full_size_training_steps_reduce_the_loss()


def _full_size_training_steps_reduce_the_loss():
    """baseline.yml at full size, B = 2, from images: three SGD steps on a fixed batch bring the loss down, every gradient
    is finite, and the weights the inference plan was folded from are refreshed when the model goes back to eval()."""
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    from fiery_amd.synthetic import make_inputs
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg)
    randomise_weights(model)
    model = model.cuda().train()
    B = 2
    image, K, E, ego = [t.cuda() for t in make_inputs(B, model.receptive_field + model.n_future, 6, image_hw=tuple(cfg.IMAGE.FINAL_DIM))]
    gen = torch.Generator().manual_seed(4)
    labels = torch.randn(B, 1 + model.n_future, 6, *model.bev_size, generator=gen).cuda()
    targets = None
    opt = torch.optim.SGD([p for n, p in model.named_parameters() if not n.startswith('encoder.')], lr=0.05)
    losses = []
    import os
    trace = os.environ.get('FIERY_TEST_TRACE') == '1'

    def mark(what):
        if trace:
            torch.cuda.synchronize()
            print('[trace]', what, flush=True)

    if trace:                                    # where in the backward pass a device fault happens: synchronise at the seams
        pool = model.pool_engine()
        for name in ('pool', 'pool_fused'):
            inner = getattr(pool, name)

            def traced(*a, _inner=inner, _name=name, **kw):
                mark(f'before {_name} forward')
                res = _inner(*a, **kw)
                mark(f'after {_name} forward')
                if res.requires_grad:
                    res.register_hook(lambda g, _n=_name: (mark(f'gradient reached {_n} output (BEV stack backward done)'), g)[1])
                for t in a[:2]:
                    if torch.is_tensor(t) and t.requires_grad:
                        t.register_hook(lambda g, _n=_name: (mark(f'{_n} backward done'), g)[1])
                return res
            setattr(pool, name, traced)
    for _ in range(3):
        opt.zero_grad()
        out = model(image, K, E, ego, labels)
        mark('forward done')
        if targets is None:
            targets = {k: torch.randn(v.shape, generator=gen).cuda() * 0.1 for k, v in out.items() if v is not None}
        loss = sum(((out[k] - t) ** 2).mean() for k, t in targets.items())
        loss.backward()
        mark('backward done')
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        assert not bad, f'non-finite gradients in {len(bad)} tensors, e.g. {bad[:12]}'
        losses.append(loss.item())
        opt.step()
        mark('optimiser step done')
    assert losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        a = model(image, K, E, ego, None, torch.zeros(B, 1, model.latent_dim, device='cuda'))
        b = model(image, K, E, ego, None, torch.zeros(B, 1, model.latent_dim, device='cuda'))
    # (bit-equal is not on offer: the pooling kernel merges runs with LDS atomics, whose order is free)
    assert all(torch.allclose(a[k], b[k], rtol=1e-4, atol=1e-4) for k in a if a[k] is not None)


@pytest.mark.parametrize('c,relu,training,padded', [(64, True, True, False), (35, True, True, True), (35, False, True, True), (6, True, True, False),
                                                    (64, True, False, False), (21, False, False, True), (256, False, True, False)])
def test_hip_batchnorm_act_matches_torch(sim, c, relu, training, padded):
    """`HipBatchNormAct` (forward, running-statistics update, all three gradients) against F.batch_norm (+ relu): dense rows,
    rows padded to a multiple of 8 (a convolution output read in place) and a plain NCHW tensor."""
    from fiery_amd.train_graph import HipBatchNormAct
    g = torch.Generator().manual_seed(c)
    n, h, w = 3, 5, 7
    base = torch.randn(n, c, h, w, generator=g) * 2.0 + 3.0
    if padded:
        store = torch.full((n, h, w, (c + 7) // 8 * 8), 7.0)
        store[..., :c] = base.permute(0, 2, 3, 1)
        x = store[..., :c].permute(0, 3, 1, 2).requires_grad_()
    else:
        x = base.clone().requires_grad_()
    x_ref = base.clone().requires_grad_()
    weight, bias = (torch.rand(c, generator=g) + 0.5).requires_grad_(), torch.randn(c, generator=g).requires_grad_()
    w_ref, b_ref = weight.detach().clone().requires_grad_(), bias.detach().clone().requires_grad_()
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = HipBatchNormAct.apply(x, weight, bias, rm, rv, training, 0.1, 1e-5, relu, sim)
    ref = F.batch_norm(x_ref, rm_ref, rv_ref, w_ref, b_ref, training, 0.1, 1e-5)
    ref = F.relu(ref) if relu else ref
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rm, rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv_ref, rtol=1e-5, atol=1e-6)
    gy = torch.randn(ref.shape, generator=g)
    got = torch.autograd.grad(y, (x, weight, bias), gy)
    want = torch.autograd.grad(ref, (x_ref, w_ref, b_ref), gy)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, b.abs().max().item())), (a - b).abs().max()


def test_hip_spatial_mean_and_its_gradient(sim):
    from fiery_amd.train_graph import HipSpatialMean
    x = torch.randn(4, 70, 9, 11, generator=torch.Generator().manual_seed(0)).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = HipSpatialMean.apply(x, sim)
    assert torch.allclose(y, x.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)
    gy = torch.randn(4, 70)
    (got,), (want,) = torch.autograd.grad(y, x, gy), torch.autograd.grad(x.mean(dim=(2, 3)), x, gy)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)


# ---- fixtures produced by the reference in train() mode (tests/golden/make_golden.py) ---------------------------------
def _replay_training_blocks(lib, device, tol):
    import os
    import numpy as np
    from fiery_amd.modules import ResidualBottleneck, SpatialGRUWeights
    from fiery_amd.train_graph import TrainGraph
    from tests import parity_report
    from tests.golden.make_golden import training_cases
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'train_blocks.npz'))
    cases = training_cases()
    graph = TrainGraph(None, lib)

    def run_gru(module, x, h):
        outs = []
        for t in range(x.shape[1]):
            h = graph.gru_cell(x[:, t], h, module)
            outs.append(h)
        return torch.stack(outs, dim=1)

    for name, module, args, fn in (
            ('bottleneck', ResidualBottleneck(64), ('x',), lambda m, x: graph.bottleneck(x, m)),
            ('bottleneck_down', ResidualBottleneck(70, 35, downsample=True), ('x',), lambda m, x: graph.bottleneck(x, m)),
            ('gru', SpatialGRUWeights(32, 64), ('x', 'h'), run_gru)):
        torch.manual_seed(0)
        randomise_weights(module)
        module = module.train().to(device)
        inputs = [cases[name][a].clone().to(device).requires_grad_() for a in args]
        y = fn(module, *inputs)
        names, params = zip(*module.named_parameters())
        grads = torch.autograd.grad(y, list(inputs) + list(params), cases[name]['gy'].to(device))
        got = {'y': y, **{f'd_input{i}': g for i, g in enumerate(grads[:len(inputs)])},
               **{'d_' + n: g for n, g in zip(names, grads[len(inputs):])},
               **{'after_' + k: v for k, v in module.state_dict().items() if k.endswith(('running_mean', 'running_var'))}}
        worst = (0.0, '')
        for key, value in got.items():
            want = torch.from_numpy(gold[f'{name}.{key}'])
            err = (value.detach().cpu() - want).abs().max().item()
            scale = max(want.abs().max().item(), 1e-3)
            worst = max(worst, (err / scale, key))
            assert err <= tol * scale, (name, key, err, scale)
        parity_report.record(f'train_block[{name}] vs reference autograd fixture', f'{len(got)} tensors, worst: {worst[1]}', worst[0], 1.0,
                             None, None, tol, 'max-abs error relative to the tensor\'s magnitude')


def test_training_blocks_equal_the_reference_autograd_fixture_on_the_simulated_kernels(sim):
    """One Bottleneck, one down-sampling Bottleneck (70 -> 35 channels, odd-sized map), one SpatialGRU over three steps:
    outputs, input gradients, every parameter gradient and the running statistics against what the reference's own modules
    produced in train() mode (fixture `train_blocks.npz`)."""
    _replay_training_blocks(sim, 'cpu', 2e-4)


@pytest.mark.gpu
def test_training_blocks_equal_the_reference_autograd_fixture(hip):
    _replay_training_blocks(hip, 'cuda', 2e-4)


@pytest.mark.gpu
def test_trainer_step_from_images_equals_the_reference_trainer_fixture(hip):
    """`TrainingModule.shared_step(batch, is_train=True)` of the reference's trainer around the reference's model, replayed on
    the MI355X (fixture `trainer_step_tiny.npz`, made by tests/golden/make_golden.py from the unmodified fiery/trainer.py):
    the same batch of camera images, the future labels the trainer fed the model, the pinned noise, drop-connect off; this
    class in train() mode - trunk, lift head, fused lift-splat and the BEV stack on the training graph's kernels - must give
    the reference's outputs, and, fed the reference losses' own d loss / d output, the reference's parameter gradients: every
    tensor's norm and seeded projection within 1 % (of the norm, or of 1e-5 of the model's largest gradient for the biases in
    front of a train-mode BatchNorm whose gradient is zero analytically)."""
    import os
    import numpy as np
    from fiery_amd.model import Fiery
    from tests import parity_report
    from tests.golden.make_golden import attach_trainer_weights, trainer_step_case
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trainer_step_tiny.npz'))
    hparams, cfg, batch, noise = trainer_step_case()
    torch.manual_seed(0)
    model = Fiery(cfg)
    attach_trainer_weights(model)                                  # (so that the seeded weights get the trainer's key order)
    randomise_weights(model)
    model.encoder.backbone._global_params.drop_connect_rate = 0.0
    model = model.cuda().train()
    dev = lambda k: batch[k].cuda()
    out = model(dev('image'), dev('intrinsics'), dev('extrinsics'), dev('future_egomotion'),
                torch.from_numpy(gold['future_distribution_inputs']).cuda(), noise=torch.from_numpy(gold['noise']).cuda())
    keys = [k for k in out if out[k] is not None]
    assert set(keys) == {k[4:] for k in gold.files if k.startswith('out_')}
    for k in keys:
        want = torch.from_numpy(gold['out_' + k])
        err, scale = (out[k].detach().cpu() - want).abs().max().item(), max(1.0, want.abs().max().item())
        parity_report.record('trainer_step_tiny vs reference trainer fixture', k, err, scale, None, None, 2e-4 * scale)
        assert err <= 2e-4 * scale, (k, err)
    torch.autograd.backward([out[k] for k in keys], [torch.from_numpy(gold['dout_' + k]).cuda() for k in keys])
    top = max(gold[f][0] for f in gold.files if f.startswith('g_'))
    g = torch.Generator().manual_seed(5)
    rows = []
    for name, p in model.named_parameters():
        if 'g_' + name not in gold.files:
            assert p.grad is None, name
            continue
        direction = torch.randn(p.shape, generator=g)                   # (drawn for every fixture tensor, in the fixture's order)
        if name in ('segmentation_weight', 'centerness_weight', 'offset_weight', 'flow_weight'):
            continue                                                    # the uncertainty weights live in the losses, not in the model's graph
        assert p.grad is not None, name
        got = np.array([p.grad.norm().item(), (p.grad.cpu() * direction).sum().item()])
        want = gold['g_' + name]
        rows.append((np.abs(got - want).max() / max(want[0], 1e-5 * top), name))
    rows.sort(reverse=True)
    parity_report.record('trainer_step_tiny vs reference trainer fixture', f'gradient norm / projection of {len(rows)} tensors, worst ({rows[0][1][-40:]})',
                         rows[0][0], 1.0, None, None, 2e-2)
    # (the worst tensor sat at 5.9e-3, 3.2e-3 and 2.6e-3 in three runs of round 4 on the MI355X - this step, from images at
    # B = 2, has atomics in the pooling tail and the weight gradients, its summation order is not reproducible - so the bound,
    # asserted and in the ledger row alike, is 2 %)
    assert len(rows) > 300 and rows[0][0] < 2e-2, rows[:6]


@pytest.mark.gpu
def test_tiny_model_training_step_equals_the_reference_autograd_fixture(hip):
    """One training step of the tiny configuration (48 x 48 BEV, B = 2, two future frames) against the reference's `Fiery` in
    train() mode (fixture `train_model_tiny.npz`): outputs, d loss / d lifted, and per parameter tensor the gradient's norm
    and its projection on a seeded direction.  The fixture carries the same numbers from a run whose input was perturbed by
    1e-6 (reported in the ledger as a reference point).  Bound: 5 % of the gradient's norm per tensor, 2 % for the median - two
    fp32 evaluations of a training step through small-batch BatchNorms agree to about a percent (see the fp64 test above), and
    a flipped ReLU gate upstream is worth a few percent."""
    import os
    import numpy as np
    from fiery_amd.model import Fiery
    from tests import parity_report
    from tests.golden.make_golden import training_loss, training_model_case
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'train_model_tiny.npz'))
    cfg, B, n_cam = training_model_case()
    torch.manual_seed(0)
    model = Fiery(cfg)
    randomise_weights(model)
    model = model.cuda().train()
    lifted, K, E, ego, labels, noise = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, B, n_cam,
                                                    with_labels=True, with_noise=True)
    leaf = lifted.clone().cuda().requires_grad_()
    out = model.bev_forward(leaf, K.cuda(), E.cuda(), ego.cuda(), labels.cuda(), noise.cuda())
    training_loss(out).backward()
    for k, v in out.items():
        if v is not None:
            want = torch.from_numpy(gold['out_' + k])
            err, scale = (v.detach().cpu() - want).abs().max().item(), max(1.0, want.abs().max().item())
            parity_report.record('train_model_tiny vs reference fixture', k, err, scale, None, None, 2e-4 * scale)
            assert err <= 2e-4 * scale, k
    got_dx, want_dx, nudged_dx = leaf.grad.cpu()[..., ::2, ::3], torch.from_numpy(gold['d_lifted_sub']), torch.from_numpy(gold['nudged_d_lifted_sub'])
    assert (got_dx - want_dx).abs().max().item() <= 5 * (nudged_dx - want_dx).abs().max().item() + 2e-2 * want_dx.abs().max().item()
    g = torch.Generator().manual_seed(5)
    loose, rows = 0, []
    for name, p in model.named_parameters():
        if p.grad is None or name.startswith('encoder.'):
            continue
        direction = torch.randn(p.shape, generator=g)
        got = np.array([p.grad.norm().item(), (p.grad.cpu() * direction).sum().item()])
        want, nudged = gold['g_' + name], gold['nudged_g_' + name]
        err = np.abs(got - want).max()
        tight = 5 * np.abs(nudged - want).max() + 2e-3 * want[0] + 1e-6
        assert err <= max(tight, 5e-2 * want[0] + 1e-5), (name, got, want)
        loose += err > tight
        rows.append(err / (want[0] + 1e-6))
    # (the reference's own fp32 step sits ~1 % (median, relative L2) from its fp64 evaluation on networks of this kind - measured by
    # test_training_step_on_the_gpu_is_as_close_to_exact_as_fp32_torch -, far above the sensitivity to a 1e-6 input change)
    assert len(rows) > 100 and float(np.median(rows)) <= 2e-2, (loose, len(rows), float(np.median(rows)))
    parity_report.record('train_model_tiny vs reference fixture', f'gradient norm / projection of {len(rows)} tensors, median rel. error',
                         float(np.median(rows)), 1.0, None, None, 2e-2, f'{loose} tensors past 5x the reference\'s own 1e-6 sensitivity')
