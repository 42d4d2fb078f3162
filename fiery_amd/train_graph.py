"""Training-mode forward of `fiery_amd.Fiery` as an autograd graph.

The inference engine (`fiery_amd.engine`) folds BatchNorm into the convolutions and fuses epilogues - neither survives
training, where BatchNorm normalises with batch statistics and every intermediate is needed again for the backward pass.
Training therefore runs the reference's data flow (fiery/models/fiery.py:130-191, 288-339 and the modules it calls) as a
PyTorch autograd graph over pixel-major (channels-last) tensors in which

* every convolution - 95 % of the step's flops - is `HipConv2d`: forward on the MFMA implicit-GEMM kernel
  (`fiery_conv_fwd`), gradient with respect to the input on the same kernel with the weights transposed and mirrored
  (the gradient zero-stuffed first for stride 2), gradient with respect to the weights on `fiery_conv_wgrad`;
  the causal (kT, kH, kW) convolutions of the temporal model are kT such 2-D convolutions on time-shifted frames;
* BatchNorm (+ReLU) in the module's mode is `HipBatchNormAct` (`fiery_bn_train_fwd / _bwd`: batch statistics, running-average
  update, deterministic two-stage sums), `nn.SyncBatchNorm` modules run `HipSyncBatchNormAct` (one all-gather forward, one
  all-reduce backward per layer);
* the element-wise half of the GRU cell is `HipGruReset` / `HipGruOut`, the decoder's x2 upsampling `HipUpsample2x` (gather
  backward), the (2, H, W) pyramid pooling per-frame plane means (`HipSpatialMean`) + a broadcast;
* voxel pooling is `ops.VoxelPool` / `ops.LiftSplat` (HIP forward and backward);
* the image trunk and the lift head upstream of the path (round 3): `lift_head` - dense convolutions on `HipConv2d` after the
  trunk's explicit 'static same' padding, depthwise ones on `HipDepthwiseConv2d` (`fiery_depthwise_conv_nhwc` in both
  directions, `fiery_depthwise_conv_wgrad_nhwc`), BatchNorm on `HipBatchNormAct`, the squeeze-and-excite means on
  `HipSpatialMean` and its two dense layers as matrix products;
* the max-pool of the skip paths and the ego-warp (`grid_sample`) run on the library's kernels in both directions since round 3
  (`HipMaxPool2x2`, `HipEgoWarp`); what is left on PyTorch-ROCm operators over the same memory:
  swish / sigmoid gates, concatenations, residual adds, the small dense layers (matrix products) - their backward comes from
  autograd.  No MIOpen convolution is left in the graph.

`Fiery.forward` dispatches here when `model.training` is set; in training mode the latent sample is drawn from the FUTURE
distribution (fiery.py:319-325), so `future_distribution_inputs` is required.  The weight holders of `fiery_amd.modules`
are read in place: parameters receive `.grad` like those of any `nn.Module`.
"""
import math
import os

import torch
import torch.nn.functional as F

from . import native
from .backbone import mbconv_geometry, same_pads
from .ops import Buf, ConvOp, identity_chan_map, round_up


# ---------------------------------------------------------------------------------------------------------------------
# convolution on the HIP kernels, differentiable
# ---------------------------------------------------------------------------------------------------------------------
def _padded_rows(t, c, cp):
    """Marks `t` - an (N, C, H, W) view of the first C channels of pixel-major rows Cp wide whose remaining channels the
    producing kernel wrote as zeros - so that `_pixel_major` may widen it again instead of copying.  Only this module's own
    operators call it, on buffers they allocated at the padded width."""
    if cp != c:
        t._fiery_padded_rows = cp
    return t


def _pixel_major(x, pad_to=8):
    """(N, C, H, W) of any layout -> contiguous (N, H, W, Cp) with Cp = C rounded up to `pad_to`.  The channels past C are
    zeros.  A tensor that one of the operators below produced as a channel slice of zero-padded rows (`_padded_rows`) is
    widened in place - no copy; the mark, the stride pattern and the row alignment of the storage offset must all agree, so
    that a slice of some other wide tensor (whose neighbouring floats are data, possibly Inf / NaN) is never mistaken for
    one.  Everything else is padded by copy."""
    n, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1)
    cp = round_up(c, pad_to)
    if cp == c:
        return t.contiguous()
    if (getattr(x, '_fiery_padded_rows', 0) == cp and min(n, h, w) > 1 and t.stride() == (h * w * cp, w * cp, cp, 1) and
            t.storage_offset() % cp == 0):
        try:
            return torch.as_strided(t, (n, h, w, cp), t.stride())
        except RuntimeError:                                  # rows that end with the storage: not padded after all
            pass
    return F.pad(t, (0, cp - c)).contiguous()


def _detached(x):
    """x.detach().float() that keeps the padded-rows mark (a Python attribute: `detach()` returns a new tensor object without
    it, and with it gone `_pixel_major` pads by copy - every forward of a 35- / 70-channel or image-trunk layer did)."""
    t = x.detach().float()
    mark = getattr(x, '_fiery_padded_rows', 0)
    if mark and t.data_ptr() == x.data_ptr():
        t._fiery_padded_rows = mark
    return t


_UNIT_EPILOGUE = {}


def _unit_epilogue(cout, device):
    """(scale = 1, shift = 0) rows of the kernel's epilogue, padded to its 32-channel tiles; one pair per width/device."""
    key = (round_up(cout, 32), str(device))
    if key not in _UNIT_EPILOGUE:
        _UNIT_EPILOGUE[key] = (torch.ones(key[0], dtype=torch.float32, device=device),
                               torch.zeros(key[0], dtype=torch.float32, device=device))
    return _UNIT_EPILOGUE[key]


# Matrix-core precision of the training graph's convolutions (forward, input gradient, and the 3 x 3 layers' weight gradient):
# 'f32' - the reference's arithmetic - or 'bf16': operands rounded to bf16 on chip, fp32 accumulation, fp32 tensors in memory -
# the library's counterpart of the reference's mixed-precision recipe (`PRECISION: 16`, fiery/configs/baseline.yml:6, trainer
# flag `precision=16` in train.py:36).  `FIERY_TRAIN_PRECISION=bf16`, or set `train_graph.CONV_PRECISION` before the step.
CONV_PRECISION = {'f32': native.PRECISION_F32, 'bf16': native.PRECISION_BF16}[os.environ.get('FIERY_TRAIN_PRECISION', 'f32')]
# Round 5: the fp32 training graph's 3 x 3 / stride-1 convolutions with 64 couts or more - forward AND input gradient (the same
# launch with transposed, mirrored weights) - run as Winograd F(2x2, 3x3) like the inference path's (csrc/conv_winograd.hip; the
# weights change every step, so their G g G^T image is packed per call: ~1 MB, a few microseconds beside a 100-200 us launch).
# No per-shape timing here (one-shot ops): the form is taken wherever it applies.  `FIERY_TRAIN_WINOGRAD=0` switches it off.
TRAIN_WINOGRAD = os.environ.get('FIERY_TRAIN_WINOGRAD', '1') != '0'
# Round 6: the 3 x 3 weight gradient runs in the SPLIT mode of its kernel (conv_grad.hip: bf16 matrix cores, every fp32 operand as
# three bf16 terms - fp32 accuracy: 6.6e-7 against 5.1e-7 relative to fp64 on 64 -> 64 at 200 x 200 x 4, launches 15-20 % shorter);
# `FIERY_TRAIN_SPLIT=0`: the fp32 matrix instruction.  The forward / input-gradient Winograd launches can run in their split form
# too (`FIERY_TRAIN_SPLIT_FORWARD=1`: another 1 % of the step) but do not by default: against fp64 the step is as close with it as
# without (tests/test_train_graph.py::test_training_step_on_the_gpu_is_as_close_to_exact_...: segmentation 6.6e-5 / 7.1e-5), yet the
# tiny trainer fixture - the reference's own fp32 CPU outputs, a train-mode BatchNorm model that amplifies any rounding change -
# sits 9.2e-4 from it instead of 3.9e-4, past that test's 2e-4 x scale bound, and a bound is not moved for 1 %.
TRAIN_SPLIT = os.environ.get('FIERY_TRAIN_SPLIT', '1') != '0'
TRAIN_SPLIT_FORWARD = os.environ.get('FIERY_TRAIN_SPLIT_FORWARD', '0') != '0'


def _train_forms():
    """The optional weight images a training convolution packs (per call: the weights change every step)."""
    if not TRAIN_WINOGRAD or CONV_PRECISION != native.PRECISION_F32:
        return ()
    return ('wsplit',) if (TRAIN_SPLIT and TRAIN_SPLIT_FORWARD) else ('wino',)


def _wgrad_precision():
    """fp32 training: the 3 x 3 weight gradient in the split mode of its kernel (fp32 accuracy on the bf16 matrix cores)."""
    return native.PRECISION_F32_SPLIT if (CONV_PRECISION == native.PRECISION_F32 and TRAIN_SPLIT) else CONV_PRECISION


def _train_form(op):
    if TRAIN_WINOGRAD and op.packed_winograd_split is not None:
        op.force_form = 'wsplit'
    elif TRAIN_WINOGRAD and op.packed_winograd is not None:
        op.force_form = 'wino'
    return op


def _launch_conv(lib, x_nhwc, weight, stride, pad):
    """Plain convolution (no bias, no activation) of a pixel-major tensor on the implicit-GEMM kernel.  The weights change
    every optimiser step, so they are packed per call (a device kernel) and no tile-height timing is done."""
    n, h, w, cp = x_nhwc.shape
    cout, cin, k, _ = weight.shape
    dev = x_nhwc.device
    scale, shift = _unit_epilogue(cout, dev)
    op = _train_form(ConvOp(lib, weight, identity_chan_map(cin), (cp // 8, 0), scale, shift, dev, stride=stride, pad=(pad, pad),
                            precision=CONV_PRECISION, tune=False, forms=_train_forms()))
    ho, wo = op.out_hw(h, w)
    out = Buf.alloc(n, ho, wo, cout, dev, zero=False)
    op([Buf(x_nhwc, n, h, w, cp)], out)
    return out.tensor                                            # (n, ho, wo, round_up(cout, 8)); padding channels are 0


class HipConv2d(torch.autograd.Function):
    """conv2d(x, weight, stride, padding) with square kernels, no bias (callers add it), on libfiery_hip.so."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, lib):
        x_nhwc = _pixel_major(_detached(x))
        cout = weight.shape[0]
        y = _launch_conv(lib, x_nhwc, weight.detach().float(), stride, pad)
        ctx.save_for_backward(x_nhwc, weight)
        ctx.meta = (x.shape, stride, pad, lib)
        return _padded_rows(y[..., :cout].permute(0, 3, 1, 2), cout, y.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, weight = ctx.saved_tensors
        (n, c, h, w), stride, pad, lib = ctx.meta
        cout, _, k, _ = weight.shape
        g = _pixel_major(gy.float())                                            # (n, ho, wo, round_up(cout, 8))
        gx = gw = None
        if ctx.needs_input_grad[1]:
            dw = lib.conv_wgrad(x_nhwc, g, cout, k, stride, pad, _wgrad_precision())  # (cout, taps, cp)
            gw = dw[:, :, :c].permute(0, 2, 1).reshape(cout, c, k, k)
            if gw.untyped_storage().data_ptr() == dw.untyped_storage().data_ptr():
                # (a 1 x 1 layer with unpadded channels: the reshape is a VIEW of the zeroed chunk `conv_wgrad` accumulated into,
                # which other gradients share - autograd must be handed a tensor that owns its storage)
                gw = gw.clone()
        if ctx.needs_input_grad[0]:
            # dL/dx = correlation of the (zero-stuffed, for stride > 1) output gradient with the transposed, mirrored
            # weights, padding k - 1 - pad: the forward kernel again
            size_h, size_w = h + 2 * pad - k + 1, w + 2 * pad - k + 1
            if stride == 1:
                gs = g
            else:
                gs = g.new_zeros(n, size_h, size_w, g.shape[-1])
                gs[:, ::stride, ::stride][:, :g.shape[1], :g.shape[2]] = g
            w_t = weight.detach().float().transpose(0, 1).flip(2, 3).contiguous()
            gx = _launch_conv(lib, gs, w_t, 1, k - 1 - pad)
            gx = _padded_rows(gx[..., :c].permute(0, 3, 1, 2), c, gx.shape[-1])
        return gx, gw, None, None, None


class HipConv2dCat(torch.autograd.Function):
    """conv2d(cat([x0, x1], 1), weight, 1, pad) without the concatenation: the kernel reads its input channels from two tensors
    (the virtual concat of `fiery_conv_desc.src[0..1]` - the GRU's [x, h], layers/temporal.py:49-62).  Stride 1.  The weight
    gradient is two launches, one per source, each filling its own channel range; the input gradient one launch over all
    channels, handed back as two channel views."""

    @staticmethod
    def forward(ctx, x0, x1, weight, pad, lib):
        a, b = _pixel_major(x0.detach().float()), _pixel_major(x1.detach().float())
        n, h, w, p0 = a.shape
        p1 = b.shape[-1]
        c0, c1 = x0.shape[1], x1.shape[1]
        cout, cin, k, _ = weight.shape
        assert cin == c0 + c1 and tuple(b.shape[:3]) == (n, h, w)
        dev = a.device
        scale, shift = _unit_epilogue(cout, dev)
        op = _train_form(ConvOp(lib, weight.detach().float(), identity_chan_map(c0) + identity_chan_map(c1, offset=p0), (p0 // 8, p1 // 8),
                                scale, shift, dev, stride=1, pad=(pad, pad), precision=CONV_PRECISION, tune=False, forms=_train_forms()))
        ho, wo = op.out_hw(h, w)
        out = Buf.alloc(n, ho, wo, cout, dev, zero=False)
        op([Buf(a, n, h, w, p0), Buf(b, n, h, w, p1)], out)
        ctx.save_for_backward(a, b, weight)
        ctx.meta = (c0, c1, h, w, pad, lib)
        y = out.tensor
        return _padded_rows(y[..., :cout].permute(0, 3, 1, 2), cout, y.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        a, b, weight = ctx.saved_tensors
        c0, c1, h, w, pad, lib = ctx.meta
        cout, _, k, _ = weight.shape
        g = _pixel_major(gy.float())
        gx0 = gx1 = gw = None
        if ctx.needs_input_grad[2]:
            d0 = lib.conv_wgrad(a, g, cout, k, 1, pad, _wgrad_precision())[:, :, :c0]          # (cout, taps, c0)
            d1 = lib.conv_wgrad(b, g, cout, k, 1, pad, _wgrad_precision())[:, :, :c1]
            gw = torch.cat([d0, d1], dim=2).permute(0, 2, 1).reshape(cout, c0 + c1, k, k)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            w_t = weight.detach().float().transpose(0, 1).flip(2, 3).contiguous()
            gx = _launch_conv(lib, g, w_t, 1, k - 1 - pad)                                  # (n, h, w, round_up(c0 + c1, 8))
            gx0 = gx[..., :c0].permute(0, 3, 1, 2)
            gx1 = gx[..., c0:c0 + c1].permute(0, 3, 1, 2)
        return gx0, gx1, gw, None, None


class HipSplitChannels(torch.autograd.Function):
    """(N, 2C, H, W) -> two (N, C, H, W) channel views of it; the two gradients meet again in ONE pixel-major tensor (autograd's
    own slice backward would zero-fill and add two full-size tensors)."""

    @staticmethod
    def forward(ctx, x):
        c = x.shape[1] // 2
        ctx.c = c
        return x[:, :c], x[:, c:2 * c]

    @staticmethod
    def backward(ctx, ga, gb):
        n, c, h, w = ga.shape
        g = torch.cat([ga.permute(0, 2, 3, 1), gb.permute(0, 2, 3, 1)], dim=3)              # (n, h, w, 2c) dense rows
        return g.permute(0, 3, 1, 2)


class HipDepthwiseConv2d(torch.autograd.Function):
    """Depthwise k x k convolution with explicit (asymmetric) zero padding - `MBConvBlock._depthwise_conv` of the image trunk
    (efficientnet-pytorch, behind fiery/models/encoder.py:58-86): forward `fiery_depthwise_conv_nhwc` (no BatchNorm, no
    activation), input gradient the same kernel on the output gradient (zero-stuffed for stride 2) with the taps mirrored and
    padding k - 1 - pad, weight gradient `fiery_depthwise_conv_wgrad_nhwc`.  weight (C, 1, k, k); pads = (top, left, bottom,
    right); C a multiple of 4 (every width of the trunk is)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pads, lib):
        n, c, h, w = x.shape
        k = weight.shape[-1]
        top, left, bottom, right = pads
        ho, wo = (h + top + bottom - k) // stride + 1, (w + left + right - k) // stride + 1
        xr = x.detach().float().permute(0, 2, 3, 1).contiguous()                  # (n, h, w, c)
        taps = weight.detach().float().reshape(c, k * k).t().contiguous()        # tap-major [k*k][c]
        out = torch.empty(n, ho, wo, c, dtype=torch.float32, device=x.device)
        lib.depthwise_conv(xr, c, n, h, w, c, taps, c, k, stride, top, left, ho, wo, None, None, native.ACT_NONE, out, c)
        ctx.save_for_backward(xr, taps)
        ctx.meta = (n, c, h, w, k, stride, top, left, ho, wo, lib)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xr, taps = ctx.saved_tensors
        n, c, h, w, k, stride, top, left, ho, wo, lib = ctx.meta
        g = gy.float().permute(0, 2, 3, 1).contiguous()                           # (n, ho, wo, c)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            dw = lib.depthwise_conv_wgrad(xr, c, n, h, w, c, g, c, ho, wo, k, stride, top, left)
            gw = dw.t().reshape(c, 1, k, k)
        if ctx.needs_input_grad[0]:
            if stride == 1:
                gs, hs, ws = g, ho, wo
            else:                                                                  # zeros between the gradient's pixels
                hs, ws = (ho - 1) * stride + 1, (wo - 1) * stride + 1
                gs = g.new_zeros(n, hs, ws, c)
                gs[:, ::stride, ::stride] = g
            flipped = taps.flip(0).contiguous()                                    # tap (ky, kx) -> (k-1-ky, k-1-kx)
            gx = torch.empty(n, h, w, c, dtype=torch.float32, device=gy.device)
            lib.depthwise_conv(gs, c, n, hs, ws, c, flipped, c, k, 1, k - 1 - top, k - 1 - left, h, w, None, None, native.ACT_NONE, gx, c)
            gx = gx.permute(0, 3, 1, 2)
        return gx, gw, None, None, None


class HipUpsample2x(torch.autograd.Function):
    """`nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)` (layers/convolutions.py:203-214): forward
    `fiery_upsample2x_add_nhwc` (no shift, no skip), backward `fiery_upsample2x_bwd_nhwc` - a gather, where the stock
    backward scatters with atomics."""

    @staticmethod
    def forward(ctx, x, lib):
        n, c, h, w = x.shape
        x_nhwc = _pixel_major(x.detach().float(), 4)
        cp = x_nhwc.shape[-1]
        out = torch.empty(n, 2 * h, 2 * w, cp, dtype=torch.float32, device=x.device)
        lib.upsample2x_add(x_nhwc, cp, n, h, w, cp, None, None, 0, out, cp)
        ctx.meta = (n, c, h, w, cp, lib)
        return _padded_rows(out[..., :c].permute(0, 3, 1, 2), c, cp)

    @staticmethod
    def backward(ctx, gy):
        n, c, h, w, cp, lib = ctx.meta
        return lib.upsample2x_bwd(_pixel_major(gy.float(), 4), n, h, w, cp)[..., :c].permute(0, 3, 1, 2), None


class HipMaxPool2x2(torch.autograd.Function):
    """`F.max_pool2d(F.pad(x, odd sizes -> even, 0), 2, 2)` - the pooled skip of a down-sampling Bottleneck
    (layers/convolutions.py:150-166): forward `fiery_maxpool2x2_nhwc`, backward `fiery_maxpool2x2_bwd_nhwc` (the gradient goes
    to the window's first maximum, ATen's tie rule; what falls on the zero padding is dropped, as `F.pad`'s backward does)."""

    @staticmethod
    def forward(ctx, x, lib):
        n, c, h, w = x.shape
        x_nhwc = _pixel_major(x.detach().float(), 4)
        cp = x_nhwc.shape[-1]
        out = torch.empty(n, (h + 1) // 2, (w + 1) // 2, cp, dtype=torch.float32, device=x.device)
        lib.maxpool2x2(x_nhwc, cp, n, h, w, cp, out, cp)
        ctx.save_for_backward(x_nhwc)
        ctx.meta = (n, c, h, w, cp, lib)
        return _padded_rows(out[..., :c].permute(0, 3, 1, 2), c, cp)

    @staticmethod
    def backward(ctx, gy):
        x_nhwc, = ctx.saved_tensors
        n, c, h, w, cp, lib = ctx.meta
        gx = lib.maxpool2x2_bwd(x_nhwc, cp, _pixel_major(gy.float(), 4), cp, n, h, w, cp)
        return _padded_rows(gx[..., :c].permute(0, 3, 1, 2), c, cp), None


class HipEgoWarp(torch.autograd.Function):
    """`cumulative_warp_features` (utils/geometry.py:225-253) of all frames at once, differentiable in the features: forward
    `fiery_bev_warp_nchw_to_nhwc` (the resampling the inference path uses, pixel-major output for the temporal model), backward
    its adjoint `fiery_bev_warp_bwd_nhwc_to_nchw` (channel planes, what the pooling backward reads).  The transforms depend on
    the ego-motion only and carry no gradient."""

    @staticmethod
    def forward(ctx, x, theta, identity, lib):
        n, c, h, w = x.shape
        cp = round_up(c, 8)
        make = torch.zeros if cp != c else torch.empty
        out = make(n, h, w, cp, dtype=torch.float32, device=x.device)
        lib.bev_warp_nchw_to_nhwc(x.detach().float().contiguous(), theta, identity, out, cp, h * w * cp)
        ctx.save_for_backward(theta)
        ctx.meta = (n, c, h, w, cp, tuple(identity), lib)
        return _padded_rows(out[..., :c].permute(0, 3, 1, 2), c, cp)

    @staticmethod
    def backward(ctx, gy):
        theta, = ctx.saved_tensors
        n, c, h, w, cp, identity, lib = ctx.meta
        g = _pixel_major(gy.float())
        return lib.bev_warp_bwd(g, cp, h * w * cp, theta, identity, n, c, h, w), None, None, None


def _rows(x):
    """(N, C, H, W) tensor -> (tensor whose first element starts pixel 0, floats between pixels): the tensor itself when its
    memory is pixel-major rows (a channels-last tensor, or a convolution output with padded rows), else a dense copy."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if min(n, h, w) > 1 and sc == 1 and sw >= c and sw % 4 == 0 and sh == w * sw and sn == h * sh:
        return x, sw
    t = x.permute(0, 2, 3, 1)
    return (t if t.is_contiguous() else t.contiguous()), c


class HipBatchNormAct(torch.autograd.Function):
    """BatchNorm (batch statistics + running-average update while training, running statistics otherwise) with an optional
    fused ReLU: `fiery_bn_train_fwd` / `fiery_bn_train_bwd`.  Output rows are padded to a multiple of 8 channels (zeros), the
    form the next convolution reads in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, batch_stats, momentum, eps, relu, lib):
        n, c, h, w = x.shape
        xr, ld = _rows(x.detach().float())
        c_store = round_up(c, 8)
        y, mean, invstd = lib.bn_train_fwd(xr, ld, n * h * w, c, None if weight is None else weight.detach(),
                                           None if bias is None else bias.detach(), running_mean, running_var, batch_stats, momentum,
                                           eps, relu, c_store)
        ctx.save_for_backward(xr, y if relu else None, mean, invstd, None if weight is None else weight.detach())
        ctx.meta = (n, c, h, w, ld, c_store, batch_stats, lib)
        return _padded_rows(y.view(n, h, w, c_store)[..., :c].permute(0, 3, 1, 2), c, c_store)

    @staticmethod
    def backward(ctx, gy):
        xr, y, mean, invstd, weight = ctx.saved_tensors
        n, c, h, w, ld, c_store, batch_stats, lib = ctx.meta
        gr, g_ld = _rows(gy.float())
        gx, dgamma, dbeta = lib.bn_train_bwd(gr, g_ld, xr, ld, y, c_store, n * h * w, c, weight, mean, invstd, batch_stats, c_store)
        return (_padded_rows(gx.view(n, h, w, c_store)[..., :c].permute(0, 3, 1, 2), c, c_store), dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, None, None, None, None, None, None, None)


class HipSyncBatchNormAct(torch.autograd.Function):
    """`nn.SyncBatchNorm` in train() mode (what `sync_batchnorm=True` of train.py:35-37 turns every BatchNorm into) with an
    optional fused ReLU: this process's mean / variance (`fiery_bn_train_stats`), one all-gather of (mean, variance, count)
    combined with Chan's formula, `fiery_bn_apply`; backward: this process's sums (`fiery_bn_train_bwd_sums`), one
    all-reduce, `fiery_bn_train_bwd_dx` with the pixel count of all processes.  The parameter gradients are the LOCAL sums,
    as torch's SyncBatchNorm returns them (DistributedDataParallel averages them afterwards)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group, lib):
        import torch.distributed as dist
        n, c, h, w = x.shape
        xr, ld = _rows(x.detach().float())
        pixels = n * h * w
        mean, var = lib.bn_train_stats(xr, ld, pixels, c)
        world = dist.get_world_size(group)
        mine = torch.cat([mean, var, torch.full((1,), float(pixels), dtype=torch.float32, device=x.device)])
        everyone = torch.empty(world * (2 * c + 1), dtype=torch.float32, device=x.device)
        dist.all_gather_into_tensor(everyone, mine, group=group)
        everyone = everyone.view(world, 2 * c + 1)
        means, variances, counts = everyone[:, :c].double(), everyone[:, c:2 * c].double(), everyone[:, 2 * c:].double()
        total = counts.sum()
        g_mean = (means * counts).sum(0) / total
        g_var = ((variances + (means - g_mean) ** 2) * counts).sum(0) / total                  # biased, over all processes
        mean, invstd = g_mean.float(), torch.rsqrt(g_var + eps).float()
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_((g_var * (total / (total - 1.0))).float(), alpha=momentum)
        c_store = round_up(c, 8)
        y = lib.bn_apply(xr, ld, pixels, c, mean, invstd, None if weight is None else weight.detach(),
                         None if bias is None else bias.detach(), relu, c_store)
        ctx.save_for_backward(xr, y if relu else None, mean, invstd, None if weight is None else weight.detach())
        ctx.meta = (n, c, h, w, ld, c_store, int(total.item()), group, lib)
        return _padded_rows(y.view(n, h, w, c_store)[..., :c].permute(0, 3, 1, 2), c, c_store)

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        xr, y, mean, invstd, weight = ctx.saved_tensors
        n, c, h, w, ld, c_store, total, group, lib = ctx.meta
        gr, g_ld = _rows(gy.float())
        pixels = n * h * w
        dgamma, dbeta = lib.bn_train_bwd_sums(gr, g_ld, xr, ld, y, c_store, pixels, c, mean, invstd)
        sums = torch.cat([dgamma, dbeta])
        dist.all_reduce(sums, group=group)
        gx = lib.bn_train_bwd_dx(gr, g_ld, xr, ld, y, c_store, pixels, c, weight, mean, invstd, sums[:c].contiguous(), sums[c:].contiguous(),
                                 total, c_store)
        return (_padded_rows(gx.view(n, h, w, c_store)[..., :c].permute(0, 3, 1, 2), c, c_store), dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, None, None, None, None, None, None, None)


def _nchw_view(rows, n, h, w, c, c_store):
    return rows.view(n, h, w, c_store)[..., :c].permute(0, 3, 1, 2)


class HipGruReset(torch.autograd.Function):
    """(1 - sigmoid(pre + bias)) * h: the state the candidate convolution of `SpatialGRU.gru_cell` sees
    (layers/temporal.py:55-58); `fiery_gru_reset_fwd / _bwd`."""

    @staticmethod
    def forward(ctx, pre, bias, h, lib):
        n, c, hh, ww = pre.shape
        pr, pre_ld = _rows(pre.detach().float())
        hr, h_ld = _rows(h.detach().float())
        c_store = round_up(c, 8)
        r, rh = lib.gru_reset_fwd(pr, pre_ld, bias.detach().float().contiguous(), hr, h_ld, n * hh * ww, c, c_store)
        ctx.save_for_backward(r, hr)
        ctx.meta = (n, c, hh, ww, h_ld, c_store, lib)
        return _nchw_view(rh, n, hh, ww, c, c_store)

    @staticmethod
    def backward(ctx, g):
        r, hr = ctx.saved_tensors
        n, c, hh, ww, h_ld, c_store, lib = ctx.meta
        gr, g_ld = _rows(g.float())
        d_pre, dh = lib.gru_reset_bwd(gr, g_ld, r, hr, h_ld, n * hh * ww, c, c_store)
        return _nchw_view(d_pre, n, hh, ww, c, c_store), d_pre[:, :c].sum(dim=0), _nchw_view(dh, n, hh, ww, c, c_store), None


class HipGruOut(torch.autograd.Function):
    """h' = (1 - u) * h + u * candidate with u = sigmoid(pre + bias) (layers/temporal.py:53-54, 60-61); `fiery_gru_out_fwd / _bwd`."""

    @staticmethod
    def forward(ctx, pre, bias, h, cand, lib):
        n, c, hh, ww = pre.shape
        pr, pre_ld = _rows(pre.detach().float())
        hr, h_ld = _rows(h.detach().float())
        cr, c_ld = _rows(cand.detach().float())
        c_store = round_up(c, 8)
        u, hn = lib.gru_out_fwd(pr, pre_ld, bias.detach().float().contiguous(), hr, h_ld, cr, c_ld, n * hh * ww, c, c_store)
        ctx.save_for_backward(u, hr, cr)
        ctx.meta = (n, c, hh, ww, h_ld, c_ld, c_store, lib)
        return _nchw_view(hn, n, hh, ww, c, c_store)

    @staticmethod
    def backward(ctx, g):
        u, hr, cr = ctx.saved_tensors
        n, c, hh, ww, h_ld, c_ld, c_store, lib = ctx.meta
        gr, g_ld = _rows(g.float())
        d_pre, dh, dcand = lib.gru_out_bwd(gr, g_ld, u, hr, h_ld, cr, c_ld, n * hh * ww, c, c_store)
        return (_nchw_view(d_pre, n, hh, ww, c, c_store), d_pre[:, :c].sum(dim=0), _nchw_view(dh, n, hh, ww, c, c_store),
                _nchw_view(dcand, n, hh, ww, c, c_store), None)


class HipSpatialMean(torch.autograd.Function):
    """(N, C, H, W) -> (N, C) means over the plane (`fiery_spatial_mean`); the gradient is a broadcast."""

    @staticmethod
    def forward(ctx, x, lib):
        n, c, h, w = x.shape
        xr, ld = _rows(x.detach().float())
        out = torch.empty(n, c, dtype=torch.float32, device=x.device)
        ws = torch.empty(n * 64 * c, dtype=torch.float32, device=x.device)          # kMeanChunks partial rows per image
        lib.spatial_mean(xr, ld, h * w * ld, n, 0, 1, h * w, c, out, ws)
        ctx.meta = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.meta
        return (g * (1.0 / (h * w))).view(n, c, 1, 1).expand(n, c, h, w), None


def _stack_frames(frames):
    """[(B, C, H, W)] * T -> (B, T, C, H, W) whose memory is (B, T, H, W, C): merged to (B T, C, H, W) it is a channels-last
    batch the pixel-major operators read in place (a plain torch.stack would hand them NCHW)."""
    return torch.stack([f.permute(0, 2, 3, 1) for f in frames], dim=1).permute(0, 1, 4, 2, 3)


class _Frames(torch.autograd.Function):
    """(B, T, C, H, W) -> its T frames (views).  Indexing does the same; the difference is the backward: ONE pixel-major stack of
    the frame gradients instead of T zero-filled (B, T, C, H, W) tensors in NCHW order, a copy into each and T - 1 additions -
    and, downstream of those, every operator converting NCHW gradients back to pixel-major rows."""

    @staticmethod
    def forward(ctx, x):
        ctx.frame = (x.shape[0], *x.shape[2:])
        return tuple(x[:, t] for t in range(x.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        like = next(g for g in grads if g is not None)
        return _stack_frames([like.new_zeros(ctx.frame) if g is None else g for g in grads])


def _frames_view(y, b, t, first=0):
    """(B T, C, H, W) channels-last images -> (B, T - first, C, H, W) from frame `first` on, reshape and slice written in
    memory order: when autograd has to materialise the gradient of this view (the zero-filled gradient of the slice; a slice
    of it that went on, with no (B T) merge of its strides), the tensor comes out pixel-major.  `y.view(b, t, c, h, w)[:, first:]`
    materialises NCHW, and every backward kernel below it then converts."""
    n, c, h, w = y.shape
    frames = y.permute(0, 2, 3, 1).reshape(b, t, h, w, c)
    return (frames[:, first:] if first else frames).permute(0, 1, 4, 2, 3)


def _cat_frames(parts):
    """The same for (B, t_i, C, H, W) pieces along the frame axis."""
    return torch.cat([p.permute(0, 1, 3, 4, 2) for p in parts], dim=1).permute(0, 1, 4, 2, 3)


def _counts_batches_once(method):
    """While a whole-path entry point runs, the BatchNorm layers' `num_batches_tracked += 1` are collected and applied in one
    multi-tensor launch at the end (108 one-element kernels per step otherwise).  Block-level methods called on their own
    count at once, as before."""
    import functools

    @functools.wraps(method)
    def entry(self, *args, **kwargs):
        if self._counters is not None:                 # nested entry (forward -> bev_forward -> bev_stack): the outermost flushes
            return method(self, *args, **kwargs)
        self._counters = {}
        try:
            return method(self, *args, **kwargs)
        finally:
            pending, self._counters = self._counters, None
            if pending:
                torch._foreach_add_([t for t, _ in pending.values()], [k for _, k in pending.values()])
    return entry


class TrainGraph:
    def __init__(self, model, lib=None, conv2d=None):
        """conv2d: the differentiable convolution `(x, weight, stride, pad, lib) -> y`; `HipConv2d.apply` unless a test
        substitutes its own statement of the operator to check the graph's wiring apart from the kernels."""
        self.m = model
        self.lib = lib or model._lib or native.get()
        self._conv = conv2d or HipConv2d.apply
        self._hip_ops = conv2d is None
        self._counters = None                         # {id: (num_batches_tracked, pending increments)} inside an entry point
        self.whole_plane_pooling_as_means = True      # False: avg_pool3d + interpolate, operator for operator as the reference
        self.hip_trunk = os.environ.get('FIERY_HIP_TRUNK', '1') != '0'      # False: the image trunk + lift head as `Encoder.lift_head` (PyTorch-ROCm / MIOpen)
        if not self.hip_trunk:
            from . import exclude_miopen_nhwc_bwd_solver
            exclude_miopen_nhwc_bwd_solver()
        # (with a substituted convolution the graph may run in fp64 / on the host: resampling then stays on torch too)
        self._upsample2x = (lambda x: HipUpsample2x.apply(x, self.lib)) if conv2d is None else (
            lambda x: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False))

    # -- primitives ---------------------------------------------------------------------------------------------------
    def conv2d(self, x, conv):
        assert conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1] and conv.groups == 1
        y = self._conv(x, conv.weight, conv.stride[0], conv.padding[0], self.lib)
        if conv.bias is not None:
            y = y + conv.bias.view(1, -1, 1, 1)
        return y

    @staticmethod
    def bn(x, norm):
        """BatchNorm in the module's own mode, as `nn.BatchNorm*.forward` runs it: batch statistics, the running-average
        update and the batch counter while training; the running statistics otherwise."""
        factor = 0.0 if norm.momentum is None else norm.momentum
        if norm.training and norm.track_running_stats and norm.num_batches_tracked is not None:
            norm.num_batches_tracked.add_(1)
            if norm.momentum is None:
                factor = 1.0 / float(norm.num_batches_tracked)
        use_batch = norm.training or (norm.running_mean is None and norm.running_var is None)
        stats = (norm.running_mean, norm.running_var) if not norm.training or norm.track_running_stats else (None, None)
        # PyTorch's own BatchNorm kernels, not MIOpen's: the convolution outputs are pixel-major views (padded rows when the
        # channel count is not a multiple of 8), and MIOpen's host code crashes on them for a single image (measured: B = 1)
        with torch.backends.cudnn.flags(enabled=False):
            return F.batch_norm(x, stats[0], stats[1], norm.weight, norm.bias, use_batch, factor, norm.eps)

    def bn_act(self, x, norm, relu):
        """BatchNorm in the module's own mode (+ ReLU): the HIP kernels, or torch's operators when the graph runs with a
        substituted convolution (tests: fp64 / host evaluation)."""
        if not self._hip_ops:
            y = self.bn(x, norm)
            return F.relu(y) if relu else y
        factor = 0.0 if norm.momentum is None else norm.momentum
        tracking = norm.track_running_stats and norm.running_mean is not None
        if norm.training and tracking and norm.num_batches_tracked is not None:
            counter = norm.num_batches_tracked
            if norm.momentum is None:                 # cumulative average: the factor needs the count now (a host read)
                counter.add_(1)
                factor = 1.0 / float(counter)
            elif self._counters is None:
                counter.add_(1)
            else:
                self._counters[id(counter)] = (counter, self._counters.get(id(counter), (counter, 0))[1] + 1)
        batch_stats = norm.training or not tracking
        update = norm.training and tracking
        if isinstance(norm, torch.nn.SyncBatchNorm) and norm.training:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                group = norm.process_group if norm.process_group is not None else dist.group.WORLD
                if dist.get_world_size(group) > 1:
                    return HipSyncBatchNormAct.apply(x, norm.weight, norm.bias, norm.running_mean if update else None,
                                                     norm.running_var if update else None, factor, norm.eps, relu, group, self.lib)
        return HipBatchNormAct.apply(x, norm.weight, norm.bias, norm.running_mean if (update or not batch_stats) else None,
                                     norm.running_var if (update or not batch_stats) else None, batch_stats, factor, norm.eps, relu,
                                     self.lib)

    def conv3d_frames(self, x, weight):
        """(B, C, T, H, W) x (Cout, Cin, kT, k, k) -> (B, Cout, T, H, W): `CausalConv3d`'s convolution (kT - 1 zero frames
        before the first, 'same' in space; fiery/layers/temporal.py:65-85) through the frames-major form below."""
        b, c, t, h, w = x.shape
        y = self._conv_frames(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), weight, b, t)
        return y.reshape(b, t, -1, h, w).permute(0, 2, 1, 3, 4)

    def _conv_frames(self, x, weight, b, t):
        """Frames-major: x (B T, C, H, W), any memory format -> (B T, Cout, H, W).  A (kT, k, k) kernel is kT 2-D
        convolutions, tap dt reading the frames kT - 1 - dt steps back (zeros before the first frame of every sample)."""
        cout, c, kt, k, _ = weight.shape
        h, w = x.shape[-2:]
        out = None
        for dt in range(kt):
            shift = kt - 1 - dt
            src = x
            if shift:
                frames = x.permute(0, 2, 3, 1).reshape(b, t, h, w, c)                       # pixel-major view
                src = torch.cat([frames.new_zeros(b, shift, h, w, c), frames[:, :t - shift]], dim=1).view(b * t, h, w, c).permute(0, 3, 1, 2)
            y = self._conv(src, weight[:, :, dt], 1, (k - 1) // 2, self.lib)
            out = y if out is None else out + y
        return out

    def _unit3d(self, x, blk, b, t):
        """conv + BatchNorm3d + ReLU of a (1, 1, 1) block (temporal.py:107-117) or a `CausalConv3d` (temporal.py:65-85) on
        frames-major activations: BatchNorm3d's statistics over (B, T, H, W) are BatchNorm2d's over (B T, H, W)."""
        return self.bn_act(self._conv_frames(x, blk.conv.weight, b, t), blk.norm, relu=True)

    def _project3d(self, x, proj, b, t):
        return self.bn_act(self._conv_frames(x, proj[0].weight, b, t), proj[1], relu=False)

    def _pyramid(self, x, pp, b, t):
        """`PyramidSpatioTemporalPooling` (temporal.py:167-215) on frames-major x -> list of (B T, C', H, W).  For the size
        the reference configures - the whole plane, two frames ((2, H, W), temporal_model.py:31) - the pooled map is one
        value per frame and channel: the mean of the frame and of the one before it (padding not counted), T + 1 of them
        with the padded step on the right, which the 1x1x1 block's BatchNorm sees before it is dropped; its 'bilinear
        upsampling' is a broadcast."""
        h, w = x.shape[-2:]
        c = x.shape[1]
        outs = []
        for size, branch in zip(pp.pool_sizes, pp.features):
            if tuple(size) == (2, h, w) and self.whole_plane_pooling_as_means:
                per_frame = (HipSpatialMean.apply(x, self.lib) if self._hip_ops else x.mean(dim=(2, 3))).view(b, t, c)
                pooled = torch.cat([per_frame[:, :1], 0.5 * (per_frame[:, :-1] + per_frame[:, 1:]), per_frame[:, -1:]], dim=1)
                y = self._unit3d(pooled.reshape(b * (t + 1), c, 1, 1), branch.conv_bn_relu, b, t + 1)
                y = y.reshape(b, t + 1, -1)[:, :t].reshape(b * t, -1, 1, 1)
                outs.append(y.expand(b * t, y.shape[1], h, w))
                continue
            x5 = x.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
            pooled = F.avg_pool3d(x5, kernel_size=size, stride=(1, *size[1:]), padding=(size[0] - 1, 0, 0), count_include_pad=False)
            tp, hp, wp = pooled.shape[2:]
            y = self._unit3d(pooled.permute(0, 2, 1, 3, 4).reshape(b * tp, c, hp, wp), branch.conv_bn_relu, b, tp)
            y = y.reshape(b, tp, -1, hp, wp)[:, :t].reshape(b * t, -1, hp, wp)
            outs.append(F.interpolate(y, (h, w), mode='bilinear', align_corners=False))
        return outs

    def temporal_block(self, x, tb, b, t):
        """fiery/layers/temporal.py:218-281 on frames-major x (B T, C, H, W)."""
        paths = [self._unit3d(self._unit3d(x, tb.convolution_paths[i][0], b, t), tb.convolution_paths[i][1], b, t) for i in range(2)]
        paths.append(self._unit3d(x, tb.convolution_paths[2], b, t))
        if tb.use_pyramid_pooling:
            paths.extend(self._pyramid(x, tb.pyramid_pooling, b, t))
        res = self._unit3d(torch.cat(paths, dim=1), tb.aggregation[0], b, t)
        if tb.projection is not None:
            x = self._project3d(x, tb.projection, b, t)
        return x + res

    def bottleneck3d(self, x, blk, b, t):
        """fiery/layers/temporal.py:120-164 on frames-major x."""
        L = blk.layers
        r = self._unit3d(self._unit3d(self._unit3d(x, L.conv_down_project, b, t), L.conv, b, t), L.conv_up_project, b, t)
        if blk.projection is not None:
            x = self._project3d(x, blk.projection, b, t)
        return r + x

    def temporal_model(self, x):
        """fiery/models/temporal_model.py:47-52; x (B, T, C, H, W) -> (B, 1, C', H, W)."""
        tm = self.m.temporal_model
        if not hasattr(tm, 'model'):
            return x[:, (tm.receptive_field - 1):]
        b, t, c, h, w = x.shape
        y = x.reshape(b * t, c, h, w).contiguous(memory_format=torch.channels_last)
        for stage in tm.model:
            y = self.temporal_block(y, stage, b, t) if hasattr(stage, 'convolution_paths') else self.bottleneck3d(y, stage, b, t)
        return _frames_view(y, b, t, tm.receptive_field - 1)

    def bottleneck(self, x, blk):
        """fiery/layers/convolutions.py:64-168 (Dropout2d(p=0) is the identity)."""
        L = blk.layers
        r = self.bn_act(self.conv2d(x, L.conv_down_project), L.abn_down_project[0], relu=True)
        r = self.bn_act(self.conv2d(r, L.conv), L.abn[0], relu=True)
        r = self.bn_act(self.conv2d(r, L.conv_up_project), L.abn_up_project[0], relu=True)
        if blk.projection is None:
            return r + x
        skip = x
        if blk.downsample:
            # odd sizes are padded first so that the pooled skip meets the strided convolution's size (convolutions.py:160-162)
            if self._hip_ops:
                skip = HipMaxPool2x2.apply(skip, self.lib)
            else:
                skip = F.max_pool2d(F.pad(skip, (0, skip.shape[-1] % 2, 0, skip.shape[-2] % 2), value=0), 2, 2)
        skip = self.bn_act(self.conv2d(skip, blk.projection.conv_skip_proj), blk.projection.bn_skip_proj, relu=False)
        return r + skip

    def gru_cell(self, x, state, gru):
        """fiery/layers/temporal.py:49-62 (note (1 - reset) * state)."""
        tilde = gru.conv_state_tilde
        hidden = state.shape[1]
        if self._hip_ops and hidden % 8 == 0 and gru.conv_update.stride[0] == 1:
            # update | reset as ONE convolution over the virtual concat [x, state] (the inference plan's form: an N = 2h GEMM, the
            # input read once, no concatenated copy); bias + gru_bias_init and both sigmoids inside the element-wise kernels
            pad = gru.conv_update.padding[0]
            w_gates = torch.cat([gru.conv_update.weight, gru.conv_reset.weight], dim=0)
            pre_u, pre_r = HipSplitChannels.apply(HipConv2dCat.apply(x, state, w_gates, pad, self.lib))
            rh = HipGruReset.apply(pre_r, gru.conv_reset.bias + gru.gru_bias_init, state, self.lib)
            proposal = self.bn_act(HipConv2dCat.apply(x, rh, tilde.conv.weight, tilde.conv.padding[0], self.lib), tilde.norm, relu=True)
            return HipGruOut.apply(pre_u, gru.conv_update.bias + gru.gru_bias_init, state, proposal, self.lib)
        xs = torch.cat([x, state], dim=1)
        update = torch.sigmoid(self.conv2d(xs, gru.conv_update) + gru.gru_bias_init)
        reset = torch.sigmoid(self.conv2d(xs, gru.conv_reset) + gru.gru_bias_init)
        proposal = self.bn_act(self.conv2d(torch.cat([x, (1.0 - reset) * state], dim=1), tilde.conv), tilde.norm, relu=True)
        return (1.0 - update) * state + update * proposal

    def future_prediction(self, x, hidden):
        """fiery/models/future_prediction.py:27-36; x (B, T, C, H, W), every GRU block starts from `hidden`."""
        fp = self.m.future_prediction
        for gru, blocks in zip(fp.spatial_grus, fp.res_blocks):
            state, outs = hidden, []
            # (the first block's input is the latent sample expanded over the future frames: one tensor used T times)
            frames = [x[:, 0]] * x.shape[1] if x.stride(1) == 0 else _Frames.apply(x)
            for frame in frames:
                state = self.gru_cell(frame, state, gru)
                outs.append(state)
            x = _stack_frames(outs)                                # (B, T, C, H, W) over pixel-major memory
            b, n, c, h, w = x.shape
            y = x.reshape(b * n, c, h, w)                          # a view: channels-last images, frames of a sample consecutive
            for blk in blocks:
                y = self.bottleneck(y, blk)
            x = _frames_view(y, b, n)
        return x

    def distribution(self, s_t, dm):
        """fiery/models/distributions.py:28-39, 52-56; s_t (B, 1, C, H, W) -> mu, log_sigma (B, 1, latent)."""
        b = s_t.shape[0]
        y = s_t[:, 0]
        for blk in dm.encoder.model:
            y = self.bottleneck(y, blk)
        y = F.adaptive_avg_pool2d(y, 1)
        conv = dm.last_conv[1]
        # a (B, C) x (C, 2 latent) product.  As a matrix product (rocBLAS), not as the 1x1 convolution it is written as
        # in the reference: MIOpen serves the backward of a convolution over a 1x1 map with igemm_bwd_gtcx35_nhwc_fp32_*,
        # which reads past the end of its operand (found with GPU guard pages)
        y = F.linear(y.flatten(1), conv.weight.flatten(1), conv.bias).view(b, 1, 2 * dm.latent_dim)
        mu, log_sigma = y[:, :, :dm.latent_dim], y[:, :, dm.latent_dim:]
        return mu, torch.clamp(log_sigma, dm.min_log_sigma, dm.max_log_sigma)

    def distribution_forward(self, present, future_distribution_inputs, noise):
        """fiery/models/fiery.py:288-339 in the model's current mode: training samples from the FUTURE distribution."""
        m = self.m
        b, s, _, h, w = present.shape
        present_mu, present_log_sigma = self.distribution(present, m.present_distribution)
        future_mu = future_log_sigma = None
        if future_distribution_inputs is not None:
            future_features = future_distribution_inputs[:, 1:].contiguous().view(b, 1, -1, h, w)
            future_mu, future_log_sigma = self.distribution(torch.cat([present, future_features.to(present.dtype)], dim=2),
                                                            m.future_distribution)
        if noise is None:
            noise = torch.randn_like(present_mu) if m.training else torch.zeros_like(present_mu)
        if m.training:
            if future_mu is None:
                raise ValueError('training mode samples the latent from the future distribution (fiery/models/fiery.py:319-325): '
                                 'pass future_distribution_inputs')
            mu, sigma = future_mu, torch.exp(future_log_sigma)
        else:
            mu, sigma = present_mu, torch.exp(present_log_sigma)
        sample = (mu + sigma * noise).view(b, s, m.latent_dim, 1, 1).expand(b, s, m.latent_dim, h, w)
        return sample, dict(present_mu=present_mu, present_log_sigma=present_log_sigma, future_mu=future_mu,
                            future_log_sigma=future_log_sigma)

    def basic_block(self, x, blk):
        """torchvision resnet BasicBlock (fiery/models/decoder.py:10-17)."""
        out = self.bn_act(self.conv2d(x, blk.conv1), blk.bn1, relu=True)
        out = self.bn_act(self.conv2d(out, blk.conv2), blk.bn2, relu=False)
        if blk.downsample is not None:
            x = self.bn_act(self.conv2d(x, blk.downsample[0]), blk.downsample[1], relu=False)
        return F.relu(out + x)

    def upsample_add(self, x, skip, up):
        """fiery/layers/convolutions.py:203-214."""
        x = self._upsample2x(x)
        return self.bn_act(self.conv2d(x, up.upsample_layer[1]), up.upsample_layer[2], relu=False) + skip

    def head(self, x, h):
        y = self.bn_act(self.conv2d(x, h[0]), h[1], relu=True)
        y = self.conv2d(y, h[3])
        return torch.sigmoid(y) if len(h) > 4 else y

    def decoder(self, x):
        """fiery/models/decoder.py:53-91; x (B, S, C, H, W)."""
        d = self.m.decoder
        b, s, c, h, w = x.shape
        x = x.reshape(b * s, c, h, w)
        skip1 = x
        x = self.bn_act(self.conv2d(x, d.first_conv), d.bn1, relu=True)
        for blk in d.layer1:
            x = self.basic_block(x, blk)
        skip2 = x
        for blk in d.layer2:
            x = self.basic_block(x, blk)
        skip3 = x
        for blk in d.layer3:
            x = self.basic_block(x, blk)
        x = self.upsample_add(x, skip3, d.up3_skip)
        x = self.upsample_add(x, skip2, d.up2_skip)
        x = self.upsample_add(x, skip1, d.up1_skip)
        out = dict(segmentation=self.head(x, d.segmentation_head), instance_center=self.head(x, d.instance_center_head),
                   instance_offset=self.head(x, d.instance_offset_head),
                   instance_flow=self.head(x, d.instance_future_head) if d.predict_future_flow else None)
        # (contiguous NCHW like the reference's: its losses `.view` these tensors)
        return {k: (None if v is None else v.contiguous().view(b, s, *v.shape[1:])) for k, v in out.items()}

    # -- the path -----------------------------------------------------------------------------------------------------
    @_counts_batches_once
    def bev_stack(self, x, future_egomotion, future_distribution_inputs=None, noise=None):
        """Everything after pooling; x (B, S, C, X, Y) pooled BEV features, future_egomotion (B, S, 6)."""
        m = self.m
        cfg = m.cfg
        if self._hip_ops and x.shape[1] > 1:
            b, s, c, h, w = x.shape
            ego = future_egomotion.detach().float().contiguous()
            theta = m._warp_transforms(ego)                     # 'host' mode: the reference's own CPU operators
            if theta is None:
                theta = self.lib.warp_params(ego, m.spatial_extent)
            x = HipEgoWarp.apply(x.reshape(b * s, c, h, w), theta.reshape(b * s, 6).contiguous(),
                                 [(i % s) == s - 1 for i in range(b * s)], self.lib).view(b, s, c, h, w)
        else:
            x = cumulative_warp_features(x.clone(), future_egomotion, 'bilinear', m.spatial_extent)
        if cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE:
            b, s, c = future_egomotion.shape
            h, w = x.shape[-2:]
            spatial = future_egomotion.view(b, s, c, 1, 1).expand(b, s, c, h, w)
            spatial = torch.cat([torch.zeros_like(spatial[:, :1]), spatial[:, :(m.receptive_field - 1)]], dim=1)
            x = torch.cat([x, spatial], dim=-3)
        states = self.temporal_model(x)
        output = {}
        if m.n_future > 0:
            # one dense pixel-major copy of the present state: it is read by both distributions, every GRU block and the decoder
            hidden = _Frames.apply(states)[0].contiguous(memory_format=torch.channels_last)
            present = hidden.unsqueeze(1)
            b, _, _, h, w = present.shape
            if cfg.PROBABILISTIC.ENABLED:
                sample, dist = self.distribution_forward(present, future_distribution_inputs, noise)
                output.update(dist)
                fut_in = sample.expand(-1, m.n_future, -1, -1, -1)
            else:
                fut_in = hidden.new_zeros(b, m.n_future, m.latent_dim, h, w)
            states_out = _cat_frames([present, self.future_prediction(fut_in, hidden)])
        else:
            states_out = states[:, -1:]
        output.update(self.decoder(states_out))
        return output

    def _pooled(self, intrinsics, extrinsics, lifted=None, depth_logits=None, features=None):
        """Voxel pooling of all frames through `BevEngine.pool` / `pool_fused` (ops.VoxelPool / ops.LiftSplat under
        autograd: HIP forward and backward) -> (B, S, C, X, Y)."""
        from .model import pack_sequence_dim
        m = self.m
        b, s, n = intrinsics.shape[:3]
        K, E = pack_sequence_dim(intrinsics), pack_sequence_dim(extrinsics)
        eng = m.pool_engine()
        with torch.no_grad():
            geometry = eng.geometry(K, E, m._camera_matrices(K, E))
        if lifted is not None:
            x = lifted.reshape(b * s, *lifted.shape[2:]).permute(0, 1, 3, 4, 5, 2)           # view: (F, n, D, h, w, C)
            bev = eng.pool(x, geometry)
        else:
            bev = eng.pool_fused(depth_logits.reshape(b * s, n, *depth_logits.shape[3:]),
                                 features.reshape(b * s, n, *features.shape[3:]), geometry)
        return bev.view(b, s, *bev.shape[1:])

    @_counts_batches_once
    def bev_forward(self, lifted, intrinsics, extrinsics, future_egomotion, future_distribution_inputs=None, noise=None,
                    depth_logits=None, features=None):
        """The hot path from the encoder's outputs (`Fiery.bev_forward`'s arguments), differentiable in them and in the
        weights."""
        rf = self.m.receptive_field
        cut = lambda t: None if t is None else t[:, :rf]
        bev = self._pooled(intrinsics[:, :rf].contiguous(), extrinsics[:, :rf].contiguous(), cut(lifted), cut(depth_logits),
                           cut(features))
        return self.bev_stack(bev, future_egomotion[:, :rf].contiguous(), future_distribution_inputs, noise)

    # -- the step before the path: image trunk + lift head under autograd (round 3) -------------------------------------------
    def _same_pad_conv(self, x, conv):
        """A convolution of the trunk with its 'static same' padding (fixed at construction, asymmetric: right / bottom get the
        odd pixel): dense ones on `HipConv2d` after an explicit pad, depthwise ones on `HipDepthwiseConv2d`."""
        left, right, top, bottom = same_pads(conv)
        if conv.groups == 1:
            if left or right or top or bottom:
                x = F.pad(x, (left, right, top, bottom))
            y = self._conv(x, conv.weight, conv.stride[0], 0, self.lib)
        else:
            assert conv.groups == conv.in_channels == conv.out_channels
            if self._hip_ops:
                y = HipDepthwiseConv2d.apply(x, conv.weight, conv.stride[0], (top, left, bottom, right), self.lib)
            else:
                y = F.conv2d(F.pad(x, (left, right, top, bottom)), conv.weight, None, conv.stride, 0, 1, conv.groups)
        return y if conv.bias is None else y + conv.bias.view(1, -1, 1, 1)

    def _mbconv(self, blk, inputs, drop_connect_rate):
        """efficientnet-pytorch `MBConvBlock.forward` (the package's published block: expansion 1x1 + BN + swish, depthwise
        k x k + BN + swish, squeeze-and-excite, projection 1x1 + BN, drop-connect + identity skip) on this graph's operators.
        The two dense layers of the squeeze-and-excite act on 1x1 maps: matrix products, not convolutions (section 9c)."""
        # x * sigmoid(x) as ONE operator each way (`F.silu`: the same function; its backward is one kernel reading x and the
        # gradient, where the product form ran sigmoid_backward, two multiplications and an addition and kept sigmoid(x) too)
        swish = F.silu
        x = inputs
        stride, cin, cout = mbconv_geometry(blk)
        if hasattr(blk, '_expand_conv'):
            x = swish(self.bn_act(self._same_pad_conv(x, blk._expand_conv), blk._bn0, relu=False))
        x = swish(self.bn_act(self._same_pad_conv(x, blk._depthwise_conv), blk._bn1, relu=False))
        gate = (HipSpatialMean.apply(x, self.lib) if self._hip_ops else x.mean(dim=(2, 3)))
        gate = swish(F.linear(gate, blk._se_reduce.weight.flatten(1), blk._se_reduce.bias))
        gate = F.linear(gate, blk._se_expand.weight.flatten(1), blk._se_expand.bias)
        x = torch.sigmoid(gate)[:, :, None, None] * x
        x = self.bn_act(self._same_pad_conv(x, blk._project_conv), blk._bn2, relu=False)
        if stride == 1 and cin == cout:
            if drop_connect_rate and blk.training:
                keep = 1.0 - drop_connect_rate
                mask = torch.floor(keep + torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device))
                x = x / keep * mask
            x = x + inputs
        return x

    @_counts_batches_once
    def trunk_endpoints(self, x):
        """`Encoder.trunk_endpoints` (fiery/models/encoder.py:58-86) on this graph's operators: (deep, shallow) levels."""
        enc = self.m.encoder
        trunk = enc.backbone
        endpoints = []
        x = F.silu(self.bn_act(self._same_pad_conv(x, trunk._conv_stem), trunk._bn0, relu=False))
        previous = x
        n_blocks = len(trunk._blocks)
        from .encoder import _LAST_BLOCK_DS8
        for idx, block in enumerate(trunk._blocks):
            rate = trunk._global_params.drop_connect_rate
            if rate:
                rate *= float(idx) / n_blocks
            x = self._mbconv(block, x, rate)
            if previous.size(2) > x.size(2):
                endpoints.append(previous)
            previous = x
            if enc.downsample == 8 and idx == _LAST_BLOCK_DS8[enc.version]:
                break
        endpoints.append(x)
        return (endpoints[4], endpoints[3]) if enc.downsample == 16 else (endpoints[3], endpoints[2])

    @_counts_batches_once
    def lift_head(self, images):
        """`Encoder.lift_head` (encoder.py:58-100) under autograd: -> (depth logits or None, context features)."""
        enc = self.m.encoder
        deep, shallow = self.trunk_endpoints(images)
        x = torch.cat([shallow, self._upsample2x(deep)], dim=1)
        conv = enc.upsampling_layer.conv
        x = self.bn_act(self.conv2d(x, conv[0]), conv[1], relu=True)
        x = self.bn_act(self.conv2d(x, conv[3]), conv[4], relu=True)
        x = self.conv2d(x, enc.depth_layer)
        if enc.use_depth_distribution:
            return x[:, :enc.D], x[:, enc.D:(enc.D + enc.C)]
        return None, x

    @_counts_batches_once
    def forward(self, image, intrinsics, extrinsics, future_egomotion, future_distribution_inputs=None, noise=None):
        """`Fiery.forward` (fiery.py:130-191) with autograd.  The image trunk and the lift head run on this graph's operators
        too (`lift_head`: every convolution - dense and depthwise -, BatchNorm, the plane means and the x2 upsampling on the
        HIP kernels in both directions; round 3 - until then they were the torch statement on MIOpen).  Their two factors go
        into the fused lift-splat kernel, then `bev_stack`."""
        m = self.m
        rf = m.receptive_field
        image = image[:, :rf].contiguous()
        b, s, n, c, h, w = image.shape
        flat = image.view(b * s * n, c, h, w)
        depth_logits, features = self.lift_head(flat) if self.hip_trunk else m.encoder.lift_head(flat)
        fh, fw = features.shape[-2:]
        feats = features.view(b, s, n, -1, fh, fw)
        if depth_logits is None:
            lifted = feats.unsqueeze(4).expand(b, s, n, feats.shape[3], m.depth_channels, fh, fw)
            return self.bev_forward(lifted, intrinsics, extrinsics, future_egomotion, future_distribution_inputs, noise)
        return self.bev_forward(None, intrinsics, extrinsics, future_egomotion, future_distribution_inputs, noise,
                                depth_logits=depth_logits.view(b, s, n, -1, fh, fw), features=feats)


# ---------------------------------------------------------------------------------------------------------------------
# ego-warp with autograd (fiery/utils/geometry.py:82-157, 181-253) - torch operators, differentiable in the features
# ---------------------------------------------------------------------------------------------------------------------
def _euler_to_matrix(angle):
    """Rx . Ry . Rz of the three angles (geometry.py:109-140)."""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zeros, ones = torch.zeros_like(x), torch.ones_like(x)
    rot_z = torch.stack([torch.cos(z), -torch.sin(z), zeros, torch.sin(z), torch.cos(z), zeros, zeros, zeros, ones], 1).view(-1, 3, 3)
    rot_y = torch.stack([torch.cos(y), zeros, torch.sin(y), zeros, ones, zeros, -torch.sin(y), zeros, torch.cos(y)], 1).view(-1, 3, 3)
    rot_x = torch.stack([ones, zeros, zeros, zeros, torch.cos(x), -torch.sin(x), zeros, torch.sin(x), torch.cos(x)], 1).view(-1, 3, 3)
    return rot_x.bmm(rot_y).bmm(rot_z)


def _pose_to_matrix(vec):
    """(..., 6) -> (..., 4, 4) (geometry.py:143-157)."""
    shape = vec.shape[:-1]
    v = vec.reshape(-1, 6)
    top = torch.cat([_euler_to_matrix(v[:, 3:]), v[:, :3].unsqueeze(-1)], dim=2)
    mat = torch.cat([top, top.new_zeros(top.shape[0], 1, 4)], dim=1)
    mat[:, 3, 3] = 1.0
    return mat.view(*shape, 4, 4)


def _warp(x, matrix, mode, spatial_extent):
    """warp_features (geometry.py:181-222) given the accumulated 4x4 pose: only (tx, ty, rz) matter."""
    angle = torch.atan2(-matrix[:, 0, 1], matrix[:, 0, 0])                       # mat2pose_vec's rz (geometry.py:82-106)
    tx, ty = matrix[:, 0, 3], matrix[:, 1, 3]
    c, s = torch.cos(angle), torch.sin(angle)
    theta = torch.stack([c, -s, ty / spatial_extent[1], s, c, -(tx / spatial_extent[0])], dim=-1).view(-1, 2, 3)
    grid = F.affine_grid(theta, size=x.shape, align_corners=False)
    return F.grid_sample(x, grid.to(x.dtype), mode=mode, padding_mode='zeros', align_corners=False)


def cumulative_warp_features(x, flow, mode, spatial_extent):
    """geometry.py:225-253: frame t is warped by flow[t] @ ... @ flow[S-2]; the last frame stays."""
    s = x.shape[1]
    if s == 1:
        return x
    mats = _pose_to_matrix(flow)
    out = [x[:, -1]]
    cum = mats[:, -2]
    for t in reversed(range(s - 1)):
        out.append(_warp(x[:, t], cum, mode, spatial_extent))
        cum = mats[:, t - 1] @ cum
    return torch.stack(out[::-1], 1)
