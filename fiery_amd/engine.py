"""Execution plan of the camera-to-BEV hot path on libfiery_hip.so.

`BevEngine` turns the weight holders of `fiery_amd.modules` into packed kernel operands once
(BatchNorm folded, 1x1 siblings fused, concatenations replaced by channel placement) and then replays
the reference's inference data flow (fiery/models/fiery.py:130-191) as a fixed list of kernel launches
on the caller's HIP stream:

  geometry -> voxel pooling -> ego-warp (+NCHW->NHWC) -> temporal blocks -> present/future distribution
  -> SpatialGRU future prediction -> decoder -> heads (NHWC->NCHW)

Work the reference computes and then discards is not computed: `TemporalModel` returns only its last
time step (fiery/models/temporal_model.py:52), so block j only evaluates output frames >= j+1.
Every intermediate lives in pixel-major (NHWC) buffers owned by this object; the library itself holds
no device memory.  Inference (`model.eval()`) only - the backward kernels are a later row of the scope
table.
"""
import os

import torch
import torch.nn as nn

from . import native
from . import ops
from .ops import HeadsOut, Buf, ConvOp, fold_bn, identity_chan_map, round_up

RELU, NONE, SIGMOID = native.ACT_RELU, native.ACT_NONE, native.ACT_SIGMOID


def _w2d(conv):
    """(Cout, Cin) matrix of a 1x1(x1) convolution."""
    return conv.weight.detach().float().reshape(conv.weight.shape[0], conv.weight.shape[1])


class _Bottleneck:
    """fiery/layers/convolutions.py:64-168 as three (four with a projected skip) fused convolutions."""

    def __init__(self, eng, mod, in_split=None):
        lib, dev = eng.lib, eng.device
        L = mod.layers
        cin, mid, cout = mod.in_channels, mod.mid_channels, mod.out_channels
        self.cin, self.mid, self.cout, self.down = cin, mid, cout, mod.downsample
        if in_split is None:
            in_split = (cin, 0)
        c0, c1 = in_split
        self.in_split = in_split
        p0 = round_up(c0, 8)
        cmap = identity_chan_map(c0) + identity_chan_map(c1, offset=p0)
        units = (p0 // 8, round_up(c1, 8) // 8)
        sc, sh = fold_bn(L.abn_down_project[0], mid)
        self.conv1 = ConvOp(lib, L.conv_down_project.weight, cmap, units, sc, sh, dev, act=RELU)
        sc, sh = fold_bn(L.abn[0], mid)
        self.conv2 = ConvOp(lib, L.conv.weight, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), sc, sh, dev,
                            stride=2 if self.down else 1, act=RELU)
        sc, sh = fold_bn(L.abn_up_project[0], cout)
        self.conv3 = ConvOp(lib, L.conv_up_project.weight, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), sc, sh,
                            dev, act=RELU)
        # 3x3 -> 1x1 up-projection (+ residual) as one kernel when the shapes allow: the mid tensor stays on chip
        self.fused_tail = None
        if mid <= 32 and cout <= 64:
            s2, b2 = fold_bn(L.abn[0], mid)
            self.fused_tail = ConvOp(lib, L.conv.weight, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), s2, b2, dev,
                                     stride=2 if self.down else 1, act=RELU).chain_pointwise(
                                         L.conv_up_project.weight, sc, sh, RELU)
        self.next_tail = None            # the fused tail that also computes the next block's down-projection (attach_next)
        self._down = (L.conv_down_project.weight, fold_bn(L.abn_down_project[0], mid))
        self.skip = None
        if mod.projection is not None:
            sc, sh = fold_bn(mod.projection.bn_skip_proj, cout)
            self.skip = ConvOp(lib, mod.projection.conv_skip_proj.weight, identity_chan_map(cin),
                               (round_up(cin, 8) // 8, 0), sc, sh, dev)

    def out_hw(self, H, W):
        return ((H + 1) // 2, (W + 1) // 2) if self.down else (H, W)

    def attach_next(self, nxt):
        """Let this block's fused tail also compute the 1x1 down-projection of `nxt`, the block that consumes its output
        (same kernel, the finished tile still on chip): `nxt` then starts from that tensor and skips its own first launch -
        a memory-bound layer that would re-read what this block has just written."""
        if (self.fused_tail is not None and self.skip is None and not self.down and self.cout == 64 and nxt.cin == 64 and
                nxt.in_split[1] == 0 and nxt.mid <= 32 and os.environ.get('FIERY_CHAIN_NEXT', '1') != '0'):
            w, (sc, sh) = nxt._down
            self.next_tail = self.fused_tail.chain_next(w, sc, sh, RELU)

    def run(self, eng, srcs, out, scratch, t1_ready=None, next_t1=None):
        """srcs: one Buf, or two for a split input; scratch: dict name -> Buf factory results.  t1_ready: this block's
        down-projection, already computed by the previous block's tail; next_t1: where to put the next block's."""
        x = srcs[0]
        n, H, W = x.n_img, x.H, x.W
        Ho, Wo = self.out_hw(H, W)
        t1 = t1_ready
        if t1 is None:
            t1 = scratch('bn_t1', n, H, W, self.mid)
            self.conv1(list(srcs), t1)
        if self.fused_tail is None:
            t2 = scratch('bn_t2', n, Ho, Wo, self.mid)
            self.conv2([t1], t2)
            tail = lambda res: self.conv3([t2], out, res=res)
        elif next_t1 is not None and self.next_tail is not None:
            tail = lambda res: self.next_tail([t1], out, res=res, out3=next_t1)
        else:
            tail = lambda res: self.fused_tail([t1], out, res=res)
        if self.skip is None:
            assert len(srcs) == 1
            tail(x)
            return
        if self.down:
            pooled = scratch('bn_pool', n, Ho, Wo, self.cin)
            off = 0
            for s, c in zip(srcs, self.in_split):
                if c:
                    eng.lib.maxpool2x2(s, s.ld, n, H, W, s.C, pooled.slice(off, s.C), pooled.ld)
                    off += s.C
            skip_in = pooled
        else:
            assert len(srcs) == 1
            skip_in = x
        s2 = scratch('bn_skip', n, Ho, Wo, self.cout)
        self.skip([skip_in], s2)
        tail(s2)


from .backbone import mbconv_geometry, same_pads as _same_pads      # noqa: E402  (layer-derived: any EfficientNet with the package's names)


def _out_size(size, k, stride, before, after):
    return (size + before + after - k) // stride + 1


class _MBConv:
    """One MBConv block of the image trunk (efficientnet-pytorch `MBConvBlock`, reached through
    fiery/models/encoder.py:58-86): 1x1 expansion + BN + swish (MFMA conv kernel), depthwise k x k + BN + swish
    (`fiery_depthwise_conv_nhwc`), squeeze-and-excite (spatial mean, two small dense layers, channel gate), 1x1 projection
    + BN (+ identity skip) (MFMA conv kernel).  Inference only: drop-connect is the identity."""

    def __init__(self, eng, blk):
        lib, dev = eng.lib, eng.device
        self.stride, self.cin, self.cout = mbconv_geometry(blk)
        dw = blk._depthwise_conv
        self.mid = dw.in_channels
        self.k = dw.kernel_size[0]
        self.pads = _same_pads(dw)
        self.expand = None
        if hasattr(blk, '_expand_conv'):
            assert _same_pads(blk._expand_conv) == (0, 0, 0, 0)
            sc, sh = fold_bn(blk._bn0, self.mid)
            self.expand = ConvOp(lib, blk._expand_conv.weight, identity_chan_map(self.cin), (round_up(self.cin, 8) // 8, 0),
                                 sc, sh, dev, pad=(0, 0), act=native.ACT_SWISH)
        wld = round_up(self.mid, 4)
        w = torch.zeros(self.k * self.k, wld, dtype=torch.float32)
        w[:, :self.mid] = dw.weight.detach().float().cpu().view(self.mid, self.k * self.k).t()
        self.dw_w, self.dw_ld = w.to(dev), wld
        sc, sh = fold_bn(blk._bn1, self.mid)
        self.dw_scale, self.dw_shift = sc.to(dev), sh.to(dev)
        self.sq = blk._se_reduce.out_channels
        self.se_w1 = blk._se_reduce.weight.detach().float().view(self.sq, self.mid).contiguous().to(dev)
        self.se_b1 = blk._se_reduce.bias.detach().float().contiguous().to(dev)
        self.se_w2 = blk._se_expand.weight.detach().float().view(self.mid, self.sq).contiguous().to(dev)
        self.se_b2 = blk._se_expand.bias.detach().float().contiguous().to(dev)
        sc, sh = fold_bn(blk._bn2, self.cout)
        self.project = ConvOp(lib, blk._project_conv.weight, identity_chan_map(self.mid), (round_up(self.mid, 8) // 8, 0),
                              sc, sh, dev, pad=(0, 0))
        self.skip = self.stride == 1 and self.cin == self.cout

    def run(self, eng, x, tag, parity):
        lib = eng.lib
        n, H, W = x.n_img, x.H, x.W
        e = x
        if self.expand is not None:
            e = eng.buf(tag + 'e', n, H, W, self.mid)
            self.expand([x], e)
        left, right, top, bottom = self.pads
        Ho, Wo = _out_size(H, self.k, self.stride, top, bottom), _out_size(W, self.k, self.stride, left, right)
        d = eng.buf(tag + 'd', n, Ho, Wo, self.mid)
        lib.depthwise_conv(e, e.ld, n, H, W, self.mid, self.dw_w, self.dw_ld, self.k, self.stride, top, left, Ho, Wo,
                           self.dw_scale, self.dw_shift, native.ACT_SWISH, d, d.ld)
        cpad = round_up(self.mid, 4)
        ws = eng.vec(tag + 'mw', n, cpad * 64)
        gate = eng.vec(tag + 'g', n, cpad)
        lib.se_gate_nhwc(d, d.ld, d.img_stride, n, Ho * Wo, self.mid, self.se_w1, self.se_b1, self.sq, self.se_w2, self.se_b2,
                         gate, cpad, ws)
        lib.scale_channels(d, d.ld, n, Ho * Wo, self.mid, gate, cpad)
        out = eng.buf(f'{tag}o{parity}', n, Ho, Wo, self.cout)
        self.project([d], out, res=x if self.skip else None)
        return out


# The temporal blocks' 35-channel paths on 32-aligned columns (64 channels, 29 of them zeros), in fp32 mode too since round 6: the
# (1, 3, 3) layers then are Winograd layers and the (2, 3, 3) ones two-source 3 x 3 launches over [frame t - 1 | frame t] - both
# in the split form: 242 + 128 us -> 136 + 80 us for the pair, +38 us on the wider sibling 1x1: the step +1.6 % (profiles/
# r6_temporal_pad32.txt).  With the fp32 instruction alone the padded layout had measured 1.5 % SLOWER.  FIERY_TEMPORAL_PAD32=0: 8-aligned.
TEMPORAL_PAD32 = os.environ.get('FIERY_TEMPORAL_PAD32', '1') != '0'


class _TemporalBlock:
    """fiery/layers/temporal.py:218-281 for the output frames that are still alive downstream."""

    def __init__(self, eng, tb, ego_channels):
        lib, dev = eng.lib, eng.device
        self.cin, self.cout, self.half = tb.in_channels, tb.out_channels, tb.half_channels
        self.ego = ego_channels
        self.cf = self.cin - ego_channels                     # channels that really vary in space
        # each path's channels start on a 32-aligned column of the combined tensors (TEMPORAL_PAD32; 8-aligned without), so that
        # the causal convolutions read whole 32-channel stages (35 -> 64 channels of which 29 are zeros: 1.8x the products, but
        # on the bf16 / split kernels instead of the per-lane addressed fp32 loop)
        # (round 6: with the split tile kernels among the fp32 candidates the padded form is measured again - FIERY_TEMPORAL_PAD32)
        hp = self.hp = round_up(self.half, 32 if (eng.precision == native.PRECISION_BF16 or TEMPORAL_PAD32) else 8)
        cf_pad = round_up(self.cf, 8)
        paths = tb.convolution_paths
        firsts = [paths[0][0], paths[1][0], paths[2]]
        # three sibling 1x1x1 convolutions as one GEMM; each path's channels start on an 8-aligned column
        wf = torch.zeros(3 * hp, self.cin)
        sc = torch.zeros(3 * hp)
        sh = torch.zeros(3 * hp)
        for i, blk in enumerate(firsts):
            wf[i * hp:i * hp + self.half] = _w2d(blk.conv).cpu()
            s_, b_ = fold_bn(blk.norm, self.half)
            sc[i * hp:i * hp + self.half], sh[i * hp:i * hp + self.half] = s_, b_
        self.fused = ConvOp(lib, wf[:, :self.cf].reshape(3 * hp, self.cf, 1, 1), identity_chan_map(self.cf),
                            (cf_pad // 8, 0), sc, sh, dev, act=RELU)
        self.fused_ego_w = wf[:, self.cf:].contiguous().to(dev) if ego_channels else None
        self.causal, self.causal2d = [], []
        for i in range(2):
            cc = paths[i][1]
            s_, b_ = fold_bn(cc.norm, self.half)
            self.causal.append(ConvOp(lib, cc.conv.weight, identity_chan_map(self.half), (hp // 8, 0), s_, b_, dev,
                                      pad=((cc.kernel_size[1] - 1) // 2, (cc.kernel_size[2] - 1) // 2), act=RELU))
            # A (2, 3, 3) causal convolution IS a 3 x 3 convolution of the concatenation [frame t - 1 | frame t]: as a two-source
            # 2D launch it is a layer the Winograd forms cover (with 32-aligned paths: whole 16-channel stages per source).
            # Output frames here are t >= 1 of the block's input, so frame t - 1 always exists (layers/temporal.py:30-36: the
            # causal front padding only ever meets the first frame, which no consumer reads).
            self.causal2d.append(None)
            if TEMPORAL_PAD32 and tuple(cc.kernel_size) == (2, 3, 3):
                w3 = cc.conv.weight.detach().float()                                            # (cout, cin, 2, 3, 3)
                w2 = torch.cat([w3[:, :, 0], w3[:, :, 1]], dim=1).contiguous()                  # (cout, 2 cin, 3, 3): [t - 1 | t]
                self.causal2d[-1] = ConvOp(lib, w2, identity_chan_map(self.half) + identity_chan_map(self.half, offset=hp),
                                           (hp // 8, hp // 8), s_, b_, dev, pad=(1, 1), act=RELU)
        agg = tb.aggregation[0]
        wagg = _w2d(agg.conv).cpu()
        cmap = ([i for i in range(self.half)] + [hp + i for i in range(self.half)] +
                [2 * hp + i for i in range(self.half)])
        s_, b_ = fold_bn(agg.norm, self.cout)
        self.agg = ConvOp(lib, wagg[:, :3 * self.half].reshape(self.cout, 3 * self.half, 1, 1), cmap,
                          (2 * hp // 8, hp // 8), s_, b_, dev, act=RELU)
        self.pool = None
        if tb.use_pyramid_pooling:
            red = tb.reduction_channels
            pconv = tb.pyramid_pooling.features[0].conv_bn_relu
            s_, b_ = fold_bn(pconv.norm, red)
            self.pool = dict(red=red, w=_w2d(pconv.conv).contiguous().to(dev), scale=s_.to(dev), shift=b_.to(dev),
                             wagg=wagg[:, 3 * self.half:].contiguous().to(dev))
        self.proj = None
        if tb.projection is not None:
            wp = _w2d(tb.projection[0]).cpu()
            s_, b_ = fold_bn(tb.projection[1], self.cout)
            self.proj = ConvOp(lib, wp[:, :self.cf].reshape(self.cout, self.cf, 1, 1), identity_chan_map(self.cf),
                               (cf_pad // 8, 0), s_, b_, dev)
            self.proj_ego_w = wp[:, self.cf:].contiguous().to(dev) if ego_channels else None

    def run(self, eng, xin, t_in0, ego_in, B, S, tag, out=None):
        """xin: frames t >= t_in0 of every batch element, images ordered (b, t).  Returns frames >= t_in0 + 1 (in `out`,
        a Buf of B * T_out images, when given: the last block writes the present state where its consumers read it)."""
        lib, dev = eng.lib, eng.device
        H, W = xin.H, xin.W
        T_in = S - t_in0
        T_out = T_in - 1
        t_out0 = t_in0 + 1
        hw_ld = H * W * xin.ld
        hp = self.hp
        # -- 1x1x1 siblings on every input frame
        P = eng.buf(tag + 'P', B * T_in, H, W, 3 * hp)
        bias = None
        if self.ego:
            bias = eng.vec(tag + 'fbias', B * T_in, self.fused.cout_pad)
            rows = ego_in[:, t_in0:].reshape(B * T_in, self.ego).contiguous()
            lib.rowwise_dense(rows, self.ego, B * T_in, self.ego, self.fused_ego_w, self.ego, 0, 3 * hp, None, None, NONE,
                              False, bias, self.fused.cout_pad)
        self.fused([xin], P, img_bias=bias)
        # -- the two causal paths, output frames only
        Q = eng.buf(tag + 'Q', B * T_out, H, W, 2 * hp)
        p_ld = H * W * P.ld
        for i, op in enumerate(self.causal):
            if self.causal2d[i] is not None:           # [frame t - 1 | frame t] of the path as the two sources of a 3 x 3 launch
                prev = P.slice(i * hp, hp)
                cur = Buf(P.tensor, prev.n_img, H, W, hp, P.ld, P.img_stride, prev.base_off + p_ld)
                self.causal2d[i]([(prev, T_in * p_ld, p_ld), (cur, T_in * p_ld, p_ld)], Q.slice(i * hp, hp), T_out=T_out, t_out0=t_out0,
                                 t_in_add=0)
                continue
            op([(P.slice(i * hp, hp), T_in * p_ld, p_ld)], Q.slice(i * hp, hp), T_out=T_out, t_out0=t_out0, t_in_add=1)
        # -- pyramid pooling branch: a per-frame vector, folded into the aggregation as a bias
        agg_bias = None
        if self.pool is not None:
            red = self.pool['red']
            mean = eng.vec(tag + 'mean', B * T_out, self.cf)
            ws = eng.vec(tag + 'meanws', B * T_out, self.cf * 64)
            # (2,H,W) average pooling with stride (1,H,W) and one frame of left padding that is excluded
            # from the count: output frame t averages input frames t-1 and t (layers/temporal.py:186-191);
            # t >= 1 here, so both exist: one pass over two adjacent frames of the (b, t) buffer.
            lib.spatial_mean(xin, xin.ld, T_in * hw_ld, B, hw_ld, T_out, 2 * H * W, self.cf, mean, ws)
            z = eng.vec(tag + 'z', B * T_out, round_up(red, 8))
            last = not self.ego
            lib.rowwise_dense(mean, mean.shape[1], B * T_out, self.cf, self.pool['w'], self.cin, 0, red,
                              self.pool['scale'] if last else None, self.pool['shift'] if last else None,
                              RELU if last else NONE, False, z, z.shape[1])
            if self.ego:
                # the ego-pose channels are constant over a frame; ATen's avg_pool3d adds the 2*H*W copies one by
                # one in fp32, which is reproduced bit for bit (fiery_sequential_window_mean, see fiery_hip.h)
                prev = ego_in[:, t_out0 - 1:S - 1].reshape(B * T_out, self.ego).contiguous()
                cur = ego_in[:, t_out0:].reshape(B * T_out, self.ego).contiguous()
                emean = eng.vec(tag + 'egomean', B * T_out, self.ego)
                lib.sequential_window_mean(prev, cur, B * T_out, self.ego, H * W, emean, self.ego)
                lib.rowwise_dense(emean, self.ego, B * T_out, self.ego, self.pool['w'], self.cin, self.cf, red,
                                  self.pool['scale'], self.pool['shift'], RELU, True, z, z.shape[1])
            agg_bias = eng.vec(tag + 'abias', B * T_out, self.agg.cout_pad)
            lib.rowwise_dense(z, z.shape[1], B * T_out, red, self.pool['wagg'], red, 0, self.cout, None, None, NONE, False,
                              agg_bias, self.agg.cout_pad)
        # -- skip connection + aggregation (x + relu(bn(agg)), layers/temporal.py:276-280)
        if out is None:
            out = eng.buf(tag + 'O', B * T_out, H, W, self.cout)
        assert out.n_img == B * T_out and out.C >= round_up(self.cout, 8)
        q_ld = H * W * Q.ld
        if self.proj is not None:
            R = eng.buf(tag + 'R', B * T_out, H, W, self.cout)
            pbias = None
            if self.ego:
                pbias = eng.vec(tag + 'pbias', B * T_out, self.proj.cout_pad)
                cur = ego_in[:, t_out0:].reshape(B * T_out, self.ego).contiguous()
                lib.rowwise_dense(cur, self.ego, B * T_out, self.ego, self.proj_ego_w, self.ego, 0, self.cout, None, None,
                                  NONE, False, pbias, self.proj.cout_pad)
            x_from_1 = Buf(xin.tensor, B * T_out, H, W, xin.C, xin.ld, xin.img_stride, xin.base_off + hw_ld)
            self.proj([(x_from_1, T_in * hw_ld, hw_ld)], R, img_bias=pbias, T_out=T_out, t_out0=t_out0, t_in_add=0)
            groups = [(0, B, R)]                               # one launch over every (b, t)
        elif T_out == 1:
            # identity skip, one live frame per batch element: a strided view of the input does it
            groups = [(0, B, Buf(xin.tensor, B, H, W, xin.C, xin.ld, T_in * hw_ld, xin.base_off + hw_ld))]
        else:
            # identity skip with several live frames: the input's (b, t) strides differ from the output's
            groups = [(b, 1, Buf(xin.tensor, T_out, H, W, xin.C, xin.ld, hw_ld, xin.base_off + (b * T_in + 1) * hw_ld))
                      for b in range(B)]
        for b0, nb, res in groups:
            n_o = nb * T_out
            q = Q.images(b0 * T_out, n_o)
            p2 = Buf(P.tensor, n_o, H, W, hp, P.ld, P.img_stride, P.base_off + (b0 * T_in + 1) * p_ld + 2 * hp)
            self.agg([(q, T_out * q_ld, q_ld), (p2, T_in * p_ld, p_ld)], out.images(b0 * T_out, n_o), res=res,
                     img_bias=agg_bias[b0 * T_out:b0 * T_out + n_o] if agg_bias is not None else None,
                     T_out=T_out, t_out0=t_out0, t_in_add=0)
        return out


def _border_tables(w_x):
    """(cout, cx, 3, 3) weights of spatially constant input channels -> (9 * cout, cx): for each border class
    (cy, cx) of a zero-padded 3x3 convolution the sum of the taps that fall inside the image
    (include/fiery_hip.h, img_bias_border)."""
    w64 = w_x.detach().double().cpu()
    rows = []
    for cy in range(3):
        kys = [1, 2] if cy == 0 else [0, 1] if cy == 2 else [0, 1, 2]
        for cx in range(3):
            kxs = [1, 2] if cx == 0 else [0, 1] if cx == 2 else [0, 1, 2]
            rows.append(w64[:, :, kys][:, :, :, kxs].sum(dim=(2, 3)))
    return torch.cat(rows, 0).float().contiguous()


class _Bottleneck3D:
    """fiery/layers/temporal.py:120-164 as it is used between two temporal blocks (temporal_model.py:33-36: kernel
    (1, 3, 3)): the frames are independent, so the block runs on the frames that are still alive as a batch of images."""

    def __init__(self, eng, mod):
        lib, dev = eng.lib, eng.device
        L = mod.layers
        down, conv, up = L.conv_down_project, L.conv, L.conv_up_project
        kt, kh, kw = conv.conv.weight.shape[2:]
        if kt != 1:
            raise NotImplementedError('Bottleneck3D with a temporal kernel extent is not part of any TemporalModel of the reference')
        cin, mid, cout = down.conv.weight.shape[1], down.conv.weight.shape[0], up.conv.weight.shape[0]
        self.cin, self.mid, self.cout = cin, mid, cout
        sc, sh = fold_bn(down.norm, mid)
        self.conv1 = ConvOp(lib, _w2d(down.conv).reshape(mid, cin, 1, 1), identity_chan_map(cin), (round_up(cin, 8) // 8, 0),
                            sc, sh, dev, act=RELU)
        w3 = conv.conv.weight.detach().float().reshape(mid, mid, kh, kw)
        s2, b2 = fold_bn(conv.norm, mid)
        s3, b3 = fold_bn(up.norm, cout)
        w_up = _w2d(up.conv).reshape(cout, mid, 1, 1)
        pad = ((kh - 1) // 2, (kw - 1) // 2)
        self.fused_tail = self.conv2 = self.conv3 = None
        if mid <= 32 and cout <= 64:
            self.fused_tail = ConvOp(lib, w3, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), s2, b2, dev, pad=pad,
                                     act=RELU).chain_pointwise(w_up, s3, b3, RELU)
        else:
            self.conv2 = ConvOp(lib, w3, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), s2, b2, dev, pad=pad, act=RELU)
            self.conv3 = ConvOp(lib, w_up, identity_chan_map(mid), (round_up(mid, 8) // 8, 0), s3, b3, dev, act=RELU)
        self.skip = None
        if mod.projection is not None:
            sc, sh = fold_bn(mod.projection[1], cout)
            self.skip = ConvOp(lib, _w2d(mod.projection[0]).reshape(cout, cin, 1, 1), identity_chan_map(cin),
                               (round_up(cin, 8) // 8, 0), sc, sh, dev)

    def run(self, eng, x, tag, out=None):
        n, H, W = x.n_img, x.H, x.W
        if out is None:
            out = eng.buf(tag + 'O', n, H, W, self.cout)
        t1 = eng.buf(tag + 't1', n, H, W, self.mid)
        self.conv1([x], t1)
        res = x
        if self.skip is not None:
            res = eng.buf(tag + 'skip', n, H, W, self.cout)
            self.skip([x], res)
        if self.fused_tail is not None:
            self.fused_tail([t1], out, res=res)
        else:
            t2 = eng.buf(tag + 't2', n, H, W, self.mid)
            self.conv2([t1], t2)
            self.conv3([t2], out, res=res)
        return out


class _Gru:
    """fiery/layers/temporal.py:10-62: update|reset as one N=2h GEMM, gate arithmetic in the epilogues.

    `const_x`: the input is the same vector at every pixel (the first SpatialGRU is fed the broadcast latent sample
    at every future step, fiery.py:316-330).  Its third of the K dimension then collapses into nine per-image bias
    rows - one per border class of the zero padding - and only the hidden-state channels go through the GEMM."""

    def __init__(self, eng, g, const_x=False):
        lib, dev = eng.lib, eng.device
        cx, ch = g.input_size, g.hidden_size
        self.cx, self.ch = cx, ch
        # each gate's rows start on a multiple of 16, so that the two halves of the padded GEMM are the two gates
        # (hidden sizes such as 70 = 64 + the six ego-pose channels of an identity temporal model)
        hp = round_up(ch, 16)

        def halves(update, reset):
            z = update.new_zeros((hp - ch,) + tuple(update.shape[1:]))
            return torch.cat([update, z, reset, z], 0)
        wg = halves(g.conv_update.weight.detach().float().cpu(), g.conv_reset.weight.detach().float().cpu())
        bg = halves(g.conv_update.bias.detach().float().cpu() + g.gru_bias_init,
                    g.conv_reset.bias.detach().float().cpu() + g.gru_bias_init)
        wt = g.conv_state_tilde.conv.weight.detach()
        sc, sh = fold_bn(g.conv_state_tilde.norm, ch)
        self.const_x = bool(const_x) and tuple(wg.shape[2:]) == (3, 3) and (2 * ch) % 32 == 0 and ch % 32 == 0
        if self.const_x:
            cmap, units = identity_chan_map(ch), (ch // 8, 0)
            self.gates_xw = _border_tables(wg[:, :cx]).to(dev)        # (9 * 2ch, cx)
            self.tilde_xw = _border_tables(wt[:, :cx]).to(dev)        # (9 * ch, cx)
            wg, wt = wg[:, cx:], wt[:, cx:]
        else:
            cxp = round_up(cx, 8)
            cmap = identity_chan_map(cx) + identity_chan_map(ch, offset=cxp)
            units = (cxp // 8, round_up(ch, 8) // 8)
        self.gates = ConvOp(lib, wg, cmap, units, torch.ones(2 * hp), bg, dev, epi=native.EPI_GRU_GATES)
        self.tilde = ConvOp(lib, wt, cmap, units, sc, sh, dev, act=RELU, epi=native.EPI_GRU_OUT)


class BevEngine:
    def __init__(self, model, lib, device, plan=True):
        """plan=False: geometry and pooling only (what the training graph uses - nothing derived from the weights)."""
        self.m, self.lib, self.device = model, lib, torch.device(device)
        self._bufs = {}
        cfg = model.cfg
        self.rf, self.nf, self.latent = model.receptive_field, model.n_future, model.latent_dim
        self.C = model.encoder_out_channels
        self.egopose = bool(cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE)
        self.probabilistic = bool(cfg.PROBABILISTIC.ENABLED) and self.nf > 0
        res = model.bev_resolution.detach().float().cpu().numpy()
        start = model.bev_start_position.detach().float().cpu().numpy()
        dim = model.bev_dimension.detach().cpu().numpy()
        origin = (start - res / res.dtype.type(2.0)).astype('float32')        # fp32, as fiery.py:236 evaluates it
        self.grid = native.make_grid(origin, res, dim)
        self.X, self.Y = int(dim[0]), int(dim[1])
        self.extent = model.spatial_extent
        self.frustum = model.frustum.detach().float().contiguous().to(self.device)
        self.pool_tile = int(os.environ.get('FIERY_POOL_TILE', '0'))      # voxels per LDS tile, 0 = library default
        self.pool_flags = 0
        # the engine's pooling workspace is zero-filled once and then only used by the library: calls skip their memset
        # (FIERY_POOL_NO_CLEAN=1: A/B runs against the memset form)
        self._pool_clean_flag = 0 if os.environ.get('FIERY_POOL_NO_CLEAN') == '1' else native.POOL_WORKSPACE_CLEAN
        self._pool_rank_flag = 0 if os.environ.get('FIERY_POOL_KEEP_RANKS') == '1' else native.POOL_NO_RANKS
        # matrix-core precision of every convolution of this plan ('f32' | 'bf16'; model.conv_precision)
        self.precision = {'f32': native.PRECISION_F32, 'bf16': native.PRECISION_BF16}[getattr(model, 'conv_precision', 'f32')]
        previous, ops.DEFAULT_PRECISION = ops.DEFAULT_PRECISION, self.precision
        try:
            if plan:
                self._build()
        finally:
            ops.DEFAULT_PRECISION = previous

    # -- plan ---------------------------------------------------------------------------------------
    def _build(self):
        m, dev, lib = self.m, self.device, self.lib
        # temporal model
        # temporal model: (kind, op, index of the temporal block) per stage of fiery/models/temporal_model.py:19-38
        self.temporal = []
        self.temporal_identity = not hasattr(m.temporal_model, 'model')
        if not self.temporal_identity:
            j = 0
            for stage in m.temporal_model.model:
                if hasattr(stage, 'convolution_paths'):
                    self.temporal.append(('block', _TemporalBlock(self, stage, 6 if (self.egopose and j == 0) else 0), j))
                    j += 1
                else:                                     # INBETWEEN_LAYERS > 0: Bottleneck3D (1, 3, 3) after block j - 1
                    self.temporal.append(('spatial', _Bottleneck3D(self, stage), j))
        state_c = m.future_pred_in_channels
        self.state_c = state_c
        # distributions
        self.present = self.future_dist = None
        if self.probabilistic:
            self.present = self._distribution_ops(m.present_distribution, (state_c, 0))
            self.future_dist = self._distribution_ops(m.future_distribution,
                                                      (state_c, self.nf * m.cfg.PROBABILISTIC.FUTURE_DIM))
        # future prediction
        if self.nf > 0:
            fp = m.future_prediction
            self.grus = [_Gru(self, g, const_x=(i == 0)) for i, g in enumerate(fp.spatial_grus)]
            self.res_blocks = [[_Bottleneck(self, b) for b in seq] for seq in fp.res_blocks]
            for seq in self.res_blocks:
                for a, b in zip(seq[:-1], seq[1:]):
                    a.attach_next(b)
        self._encoder_ops_built = False          # image trunk + lift head: planned on first use (_build_encoder_ops)
        # decoder
        d = m.decoder
        sc, sh = fold_bn(d.bn1, 64)
        cin = d.in_channels
        self.dec_first = ConvOp(lib, d.first_conv.weight, identity_chan_map(cin), (round_up(cin, 8) // 8, 0), sc, sh, dev,
                                stride=2, act=RELU)
        self.dec_layers = []
        for layer in (d.layer1, d.layer2, d.layer3):
            blocks = []
            for blk in layer:
                ci, co = blk.conv1.in_channels, blk.conv1.out_channels
                s1, b1 = fold_bn(blk.bn1, co)
                s2, b2 = fold_bn(blk.bn2, co)
                ops = dict(conv1=ConvOp(lib, blk.conv1.weight, identity_chan_map(ci), (ci // 8, 0), s1, b1, dev,
                                        stride=blk.stride, act=RELU),
                           conv2=ConvOp(lib, blk.conv2.weight, identity_chan_map(co), (co // 8, 0), s2, b2, dev, act=RELU,
                                        res_before_act=True),
                           down=None, cout=co, stride=blk.stride)
                if blk.downsample is not None:
                    sd, bd = fold_bn(blk.downsample[1], co)
                    ops['down'] = ConvOp(lib, blk.downsample[0].weight, identity_chan_map(ci), (ci // 8, 0), sd, bd, dev,
                                         stride=blk.stride, pad=(0, 0))
                blocks.append(ops)
            self.dec_layers.append(blocks)
        self.dec_ups = []
        for up in (d.up3_skip, d.up2_skip, d.up1_skip):
            conv, bn = up.upsample_layer[1], up.upsample_layer[2]
            ci, co = conv.in_channels, conv.out_channels
            sc, sh = fold_bn(bn, co)
            # the 1x1 convolution and the BN scale commute with bilinear interpolation (its weights sum to
            # one): run them at the low resolution, add the BN shift after upsampling
            self.dec_ups.append(dict(conv=ConvOp(lib, conv.weight, identity_chan_map(ci), (ci // 8, 0), sc,
                                                 torch.zeros(co), dev), shift=sh.to(dev), cout=co))
        heads = [('segmentation', d.segmentation_head), ('instance_center', d.instance_center_head),
                 ('instance_offset', d.instance_offset_head)]
        if d.predict_future_flow:
            heads.append(('instance_flow', d.instance_future_head))
        self.head_names = [n for n, _ in heads]
        # every head's hidden channels start on a multiple of 8 of the combined GEMM's outputs (cin is 64 in every shipped
        # configuration; 70 with an identity temporal model that carries the ego-pose channels)
        cp = round_up(cin, 8)

        def rows(t):
            return torch.cat([t.detach().float().cpu(), t.new_zeros((cp - cin,) + tuple(t.shape[1:])).float().cpu()], 0)
        wh = torch.cat([rows(h[0].weight) for _, h in heads], 0)
        scs, shs = zip(*[[rows(v) for v in fold_bn(h[1], cin)] for _, h in heads])
        self.heads_conv = ConvOp(lib, wh, identity_chan_map(cin), (round_up(cin, 8) // 8, 0), torch.cat(scs), torch.cat(shs),
                                 dev, act=RELU)
        self.heads_final = []
        for i, (name, h) in enumerate(heads):
            n_out = h[3].out_channels
            self.heads_final.append(dict(name=name, w=_w2d(h[3]).contiguous().to(dev),
                                         b=h[3].bias.detach().float().contiguous().to(dev), n_out=n_out,
                                         sigmoid=len(h) > 4, c_off=i * cp))
        # With 64 hidden channels per head (the reference's shared_out_channels) the final 1x1s ride in the 3x3 GEMM's
        # epilogue: the (images, 200, 200, 256) hidden tensor is never written.
        n_rows = sum(hd['n_out'] for hd in self.heads_final)
        self.heads_fused = cin == 64 and (len(heads) * cin) % 128 == 0 and n_rows <= native.MAX_HEAD_OUTPUTS
        if os.environ.get('FIERY_HEADS_FUSED') == '0':                 # tuning / A-B runs
            self.heads_fused = False
        if self.heads_fused:
            groups = [i for i, hd in enumerate(self.heads_final) for _ in range(hd['n_out'])]
            self.heads_conv.attach_heads(torch.cat([hd['w'] for hd in self.heads_final]),
                                         torch.cat([hd['b'] for hd in self.heads_final]), groups,
                                         [self.heads_final[g]['sigmoid'] for g in groups])
        self.head_c = cin

    def _build_encoder_ops(self):
        """Plan of the image trunk and the lift head; built on first use, because only the engine that serves
        `Fiery.forward` from images needs it (the per-sample engines of the hot path do not)."""
        if self._encoder_ops_built:
            return
        m, dev, lib = self.m, self.device, self.lib
        previous, ops.DEFAULT_PRECISION = ops.DEFAULT_PRECISION, self.precision
        try:
            self._plan_encoder_ops(m, dev, lib)
        finally:
            ops.DEFAULT_PRECISION = previous
        self._encoder_ops_built = True

    def _plan_encoder_ops(self, m, dev, lib):
        # lift head (reference: fiery/models/encoder.py:87-104, fiery/layers/convolutions.py:171-200): the coarse level is
        # interpolated into a buffer of its own and the first 3x3 reads [shallow, deep] as a virtual concat
        enc = m.encoder
        conv = enc.upsampling_layer.conv
        cs, cd, cf = enc.c_shallow, enc.c_deep, conv[0].out_channels
        ps = round_up(cs, 8)
        sc, sh = fold_bn(conv[1], cf)
        self.lh_conv1 = ConvOp(lib, conv[0].weight, identity_chan_map(cs) + identity_chan_map(cd, offset=ps),
                               (ps // 8, round_up(cd, 8) // 8), sc, sh, dev, act=RELU)
        sc, sh = fold_bn(conv[4], cf)
        self.lh_conv2 = ConvOp(lib, conv[3].weight, identity_chan_map(cf), (round_up(cf, 8) // 8, 0), sc, sh, dev, act=RELU)
        ho = enc.depth_layer.out_channels
        sc, sh = fold_bn(None, ho, enc.depth_layer.bias)
        self.lh_conv3 = ConvOp(lib, enc.depth_layer.weight, identity_chan_map(cf), (round_up(cf, 8) // 8, 0), sc, sh, dev)
        self.lh_channels = (cs, cd, cf, ho)
        # image trunk (efficientnet-pytorch through encoder.py:58-86): stem + the MBConv blocks the lift head keeps
        bb = enc.backbone
        stem = bb._conv_stem
        sc, sh = fold_bn(bb._bn0, stem.out_channels)
        self.stem_pads = _same_pads(stem)
        self.stem = ConvOp(lib, stem.weight, identity_chan_map(stem.in_channels), (round_up(stem.in_channels, 8) // 8, 0), sc, sh,
                           dev, stride=stem.stride[0], pad=(self.stem_pads[2], self.stem_pads[0]), act=native.ACT_SWISH)
        self.stem_k = stem.kernel_size[0]
        self.mbconv = [_MBConv(self, blk) for blk in bb._blocks]
        self._encoder_ops_built = True

    def _distribution_ops(self, dm, in_split):
        blocks = []
        split = in_split
        for b in dm.encoder.model:
            blocks.append(_Bottleneck(self, b, in_split=split))
            split = None
        conv = dm.last_conv[1]
        return dict(blocks=blocks, w=_w2d(conv).contiguous().to(self.device),
                    b=conv.bias.detach().float().contiguous().to(self.device),
                    lo=float(dm.min_log_sigma), hi=float(dm.max_log_sigma), compress=dm.compress_dim)

    # -- buffers ------------------------------------------------------------------------------------
    def buf(self, name, n_img, H, W, C):
        key = (name, n_img, H, W, round_up(C, 8))
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = Buf.alloc(n_img, H, W, C, self.device)
        return b

    def vec(self, name, rows, cols):
        key = (name, rows, cols)
        v = self._bufs.get(key)
        if v is None:
            v = self._bufs[key] = torch.zeros(rows, cols, dtype=torch.float32, device=self.device)
        return v

    def _scratch(self, prefix):
        return lambda name, n, H, W, C: self.buf(prefix + name, n, H, W, C)

    # -- stages -------------------------------------------------------------------------------------
    def geometry(self, intrinsics, extrinsics, camera_matrices=None):
        """`get_geometry`: (F, n, 3, 3), (F, n, 4, 4) -> (F, n, D, fH, fW, 3).  `camera_matrices` (F*n, 12) = the nine
        entries of R.K^-1 and the translation, computed by the caller (`host_camera_matrices`): the device then only
        evaluates the per-point product, which is bit-exact for any K."""
        f, n = intrinsics.shape[:2]
        if camera_matrices is None:
            cam = self.lib.camera_matrices(intrinsics.reshape(-1, 3, 3).float().contiguous(),
                                           extrinsics.reshape(-1, 4, 4).float().contiguous())
        else:
            cam = camera_matrices.to(device=self.device, dtype=torch.float32).reshape(f * n, 12).contiguous()
        geo = self.lib.lift_geometry(self.frustum, cam)
        return geo.view(f, n, *geo.shape[1:])

    def trunk_endpoints(self, image):
        """`Encoder.get_features` up to its two pyramid levels (encoder.py:58-86): (n, 3, H, W) images -> (deep, shallow)
        as pixel-major buffers (ops.Buf), the coarse level at half the resolution of the fine one."""
        self._build_encoder_ops()
        lib, enc = self.lib, self.m.encoder
        n, c, H, W = image.shape
        x0 = self.buf('tr_in', n, H, W, c)
        lib.nchw_to_nhwc(image.float().contiguous(), n, c, H * W, x0.tensor, x0.ld, x0.img_stride)
        left, right, top, bottom = self.stem_pads
        s = self.stem.stride
        x = self.buf('tr_stem', n, _out_size(H, self.stem_k, s, top, bottom), _out_size(W, self.stem_k, s, left, right),
                     self.stem.cout)
        self.stem([x0], x)
        endpoints, previous = [], x
        for idx, blk in enumerate(self.mbconv):
            # buffers are shared by role (a shape gets one expansion and one depthwise buffer; block outputs alternate
            # between two, so a block's input - its residual - and the stage's last output - an endpoint - stay intact)
            x = blk.run(self, x, 'tr_', idx % 2)
            if previous.H > x.H:
                endpoints.append(previous)
            previous = x
        endpoints.append(x)
        return (endpoints[4], endpoints[3]) if enc.downsample == 16 else (endpoints[3], endpoints[2])

    def lift_head(self, deep, shallow, out=None):
        """`Encoder.forward` after the trunk (encoder.py:87-100): deep (n, cd, h/2, w/2) and shallow (n, cs, h, w) trunk
        levels -> (depth logits (n, D, h, w) or None, context features (n, C, h, w)), planar like the trunk's tensors
        because the splat kernels read them that way.  `out` = (logits or None, features): tensors to fill instead of
        fresh ones (a caller that runs several engines on side streams allocates the results on its own stream)."""
        self._build_encoder_ops()
        lib = self.lib
        cs, cd, cf, ho = self.lh_channels
        if isinstance(deep, Buf):                                   # straight from `trunk_endpoints`
            coarse, fine = deep, shallow
            n, hd, wd, h, w = coarse.n_img, coarse.H, coarse.W, fine.H, fine.W
            assert (h, w) == (2 * hd, 2 * wd) and coarse.C == round_up(cd, 8) and fine.C == round_up(cs, 8)
        else:
            n, _, hd, wd = deep.shape
            h, w = shallow.shape[-2:]
            assert (h, w) == (2 * hd, 2 * wd) and deep.shape[1] == cd and shallow.shape[1] == cs
            fine = self.buf('lh_shallow', n, h, w, cs)
            lib.nchw_to_nhwc(shallow.float().contiguous(), n, cs, h * w, fine.tensor, fine.ld, fine.img_stride)
            coarse = self.buf('lh_deep_lo', n, hd, wd, cd)
            lib.nchw_to_nhwc(deep.float().contiguous(), n, cd, hd * wd, coarse.tensor, coarse.ld, coarse.img_stride)
        up = self.buf('lh_deep', n, h, w, cd)
        lib.upsample2x_add(coarse, coarse.ld, n, hd, wd, coarse.C, None, None, 0, up, up.ld)
        t1 = self.buf('lh_t1', n, h, w, cf)
        self.lh_conv1([fine, up], t1)
        t2 = self.buf('lh_t2', n, h, w, cf)
        self.lh_conv2([t1], t2)
        head = self.buf('lh_out', n, h, w, ho)
        self.lh_conv3([t2], head)
        D = self.m.depth_channels if self.m.encoder.use_depth_distribution else 0
        C = self.C

        def planar(c_off, c, dst):
            if c_off % 4 == 0:
                if dst is None:
                    dst = torch.empty(n, c, h, w, dtype=torch.float32, device=self.device)
                part = head.slice(c_off, c)
                lib.nhwc_to_nchw(part, part.ld, part.img_stride, n, c, h * w, dst)
                return dst
            res = head.nhwc()[..., c_off:c_off + c].permute(0, 3, 1, 2)
            if dst is None:
                return res.contiguous()
            dst.copy_(res)
            return dst
        o_logits, o_feats = out if out is not None else (None, None)
        return (planar(0, D, o_logits) if D else None), planar(D, C, o_feats)

    def _pool_workspace(self, f, n, d, h, w, device, form='compact'):
        # (one workspace per FORM: the fused lift-splat lays its tile lists out with a different plan than the compact
        # pooling, and POOL_WORKSPACE_CLEAN is a promise about the plan that used the workspace last)
        key = ('poolws', form, f, n, d, h, w, self.pool_tile, self.pool_flags)
        ws = self._bufs.get(key)
        if ws is None:
            # zero-filled once and then only ever handed to the library's pooling calls, each of which leaves it clean: they
            # run with POOL_WORKSPACE_CLEAN and skip their memset dispatch (a call that raises drops the workspace)
            ws = self._bufs[key] = self.lib.pool_workspace(f, n, d, h, w, device, self.grid, self.pool_tile, self.pool_flags,
                                                           zeroed=True)
        return ws

    def _pool_call(self, key_dims, fn, form='compact'):
        try:
            return fn()
        except Exception:
            self._bufs.pop(('poolws', form) + key_dims + (self.pool_tile, self.pool_flags), None)
            raise

    def pool(self, x, geometry, out=None):
        """`projection_to_birds_eye_view`: x logical (F, n, D, h, w, C) of any strides -> (F, C, X, Y) (written into
        `out` when given).  Differentiable with respect to x (ops.VoxelPool) when autograd is recording."""
        f, n, d, h, w, c = x.shape
        if torch.is_grad_enabled() and x.requires_grad:
            res = ops.VoxelPool.apply(x, geometry, self)
            return res if out is None else out.copy_(res)
        ws = self._pool_workspace(f, n, d, h, w, x.device)
        # The op's algorithmic bytes (4.C.N_kept + 12.N + 4.C.X.Y per frame) need N_kept, the number of
        # in-grid points: the profiling consumer counts them from the geometry after the run (`pool_algorithmic_bytes`).
        # No backward follows this call, so it leaves no voxel ranks behind (POOL_NO_RANKS: 17 MB of stores less).
        geometry = geometry.contiguous()
        detail = dict(lib=self.lib, geometry=geometry, grid=self.grid, points=f * n * d * h * w, channels=c, frames=f,
                      voxels=self.X * self.Y) if ops.PROFILE_SINK is not None else None
        return ops.profiled('voxel_pool', None, x, lambda: self._pool_call((f, n, d, h, w), lambda: self.lib.voxel_pool(
            x, x.stride(), geometry, f, n, d, h, w, c, self.grid, out=out, workspace=ws,
            tile_voxels=self.pool_tile, flags=self.pool_flags | self._pool_clean_flag | self._pool_rank_flag)), detail=detail)

    def pool_fused(self, depth_logits, features, geometry, out=None):
        """depth logits (F, n, D, h, w) + features (F, n, C, h, w) -> (F, C, X, Y) without the outer product."""
        f, n, d, h, w = depth_logits.shape
        c = features.shape[2]
        if torch.is_grad_enabled() and (depth_logits.requires_grad or features.requires_grad):
            res = ops.LiftSplat.apply(depth_logits, features, geometry, self)
            return res if out is None else out.copy_(res)
        prob = self.lib.depth_softmax(depth_logits.reshape(f * n, d, h, w).contiguous())
        ws = self._pool_workspace(f, n, d, h, w, features.device, form='fused')
        return self._pool_call((f, n, d, h, w), lambda: self.lib.lift_splat(
            prob, features.contiguous(), geometry.contiguous(), f, n, d, h, w, c, self.grid, out=out, workspace=ws,
            tile_voxels=self.pool_tile, flags=self.pool_flags | self._pool_clean_flag), form='fused')

    def _run_distribution(self, ops, srcs, tag, mu=None, log_sigma=None):
        lib = self.lib
        x = list(srcs)
        n = x[0].n_img
        for i, blk in enumerate(ops['blocks']):
            Ho, Wo = blk.out_hw(x[0].H, x[0].W)
            out = self.buf(f'{tag}d{i}', n, Ho, Wo, blk.cout)
            blk.run(self, x, out, self._scratch(f'{tag}d{i}'))
            x = [out]
        enc = x[0]
        cd = ops['compress']
        gap = self.vec(tag + 'gap', n, cd)
        ws = self.vec(tag + 'gapws', n, cd * 64)
        lib.spatial_mean(enc, enc.ld, enc.img_stride, n, 0, 1, enc.H * enc.W, cd, gap, ws)
        L = self.latent
        if mu is None:
            mu = torch.empty(n, 1, L, dtype=torch.float32, device=self.device)
        if log_sigma is None:
            log_sigma = torch.empty(n, 1, L, dtype=torch.float32, device=self.device)
        lib.rowwise_dense(gap, gap.shape[1], n, cd, ops['w'], cd, 0, L, None, ops['b'], NONE, False, mu, L)
        lib.rowwise_dense(gap, gap.shape[1], n, cd, ops['w'][L:], cd, 0, L, None, ops['b'][L:], NONE, False, log_sigma, L,
                          lo=ops['lo'], hi=ops['hi'])
        return mu, log_sigma

    def bev_stack(self, bev, future_egomotion, future_distribution_inputs=None, noise=None, into=None, theta=None):
        """Everything after pooling.  bev: (B*S, C, X, Y) NCHW; future_egomotion (B, S, 6).  `into`: optional dict of
        preallocated output tensors (by output name) to write instead of allocating.  `theta`: optional (B, S, 6) sampling
        transforms computed by the caller (`model.host_warp_transforms`) instead of by `fiery_warp_params`."""
        into = into or {}
        lib, dev = self.lib, self.device
        S = self.rf
        B = bev.shape[0] // S
        H, W, C = self.X, self.Y, self.C
        out = {}
        ego = future_egomotion.float().contiguous()
        theta_in = theta
        # -- ego-warp + layout change ---------------------------------------------------------------
        # (identity temporal model + INPUT_EGOPOSE: the state is the warped frame with the ego-pose channels behind it)
        x0 = self.buf('x0', B * S, H, W, C + (6 if (self.temporal_identity and self.egopose) else 0))
        theta = self.vec('warp_theta', B * S, 6)
        ego_in = self.vec('ego_in', B * S, 6).view(B, S, 6) if self.egopose else None          # fiery.py:152-154
        lib.warp_params(ego, self.extent, theta=theta.view(B, S, 6), ego_shifted=ego_in)
        if theta_in is not None:
            theta.view(B, S, 6).copy_(theta_in)
        identity = [(i % S) == S - 1 for i in range(B * S)]
        lib.bev_warp_nchw_to_nhwc(bev.contiguous(), theta, identity, x0.tensor, x0.ld, x0.img_stride)
        # the decoder's input holds (present, future 1 .. nf) per batch element; the temporal model's last block writes the
        # present state straight into slot 0 (no copy), and everything that reads "present" reads it there
        dec_in = present_slot = None
        if self.nf > 0:
            dec_in = self.buf('dec_in', B * (self.nf + 1), H, W, self.state_c)
            present_slot = dec_in.images(0, B, step=self.nf + 1)
        # -- temporal model --------------------------------------------------------------------------
        if self.temporal_identity:
            # fiery/models/temporal_model.py:55-62: the last frame, ego-pose channels included when INPUT_EGOPOSE is set
            # (fiery.py:147-154: frame t carries future_egomotion[t - 1], frame 0 zeros)
            present = x0.images(S - 1, B, step=S) if S > 1 else x0
            if self.egopose:
                rows = ego_in[:, S - 1].contiguous()
                lib.broadcast(rows, 6, B, H * W, 6, present.slice(C, 8), present.ld, present.img_stride)
            if present_slot is None and S > 1:            # the decoder reads its input as a dense batch of images
                present_slot = self.buf('present_dense', B, H, W, self.state_c)
            if present_slot is not None:                  # (no shipped configuration: identity temporal model over several frames)
                present_slot.nhwc().copy_(present.nhwc())
                present = present_slot
        else:
            x = x0
            n_stages = len(self.temporal)
            for i, (kind, op, j) in enumerate(self.temporal):
                final = i == n_stages - 1
                if kind == 'block':
                    x = op.run(self, x, j, ego_in, B, S, f't{j}',
                               out=present_slot if (final and S - j - 1 == 1) else None)
                else:
                    x = op.run(self, x, f's{i}', out=present_slot if (final and x.n_img == B) else None)
            present = x                                   # (B images) = the last frame
        # -- distributions ---------------------------------------------------------------------------
        if self.nf > 0:
            if self.probabilistic:
                mu, log_sigma = self._run_distribution(self.present, [present], 'pd', into.get('present_mu'),
                                                       into.get('present_log_sigma'))
                fmu = flog = None
                if future_distribution_inputs is not None:
                    lab = future_distribution_inputs[:, 1:].float().contiguous()
                    lc = lab.shape[1] * lab.shape[2]
                    labels = self.buf('labels', B, H, W, lc)
                    lib.nchw_to_nhwc(lab.view(B, lc, H * W), B, lc, H * W, labels.tensor, labels.ld, labels.img_stride)
                    fmu, flog = self._run_distribution(self.future_dist, [present, labels], 'fd', into.get('future_mu'),
                                                       into.get('future_log_sigma'))
                out.update(present_mu=mu, present_log_sigma=log_sigma, future_mu=fmu, future_log_sigma=flog)
                sample = self.vec('sample', B, self.latent)
                nz = noise.float().contiguous().view(B, self.latent) if noise is not None else None
                lib.latent_sample(mu, log_sigma, nz, self.latent, B, self.latent, sample, self.latent)
            else:
                sample = self.vec('zero_sample', B, self.latent)     # zeros (fiery.py:175-176)
            gx = None
            if not self.grus[0].const_x:                            # the broadcast input is only materialised if needed
                gx = self.buf('gru_x0', B, H, W, self.latent)
                lib.broadcast(sample, self.latent, B, H * W, self.latent, gx.tensor, gx.ld, gx.img_stride)
            if present.tensor is not dec_in.tensor:       # (a temporal model with several live output frames)
                present_slot.nhwc().copy_(present.nhwc())
                present = present_slot
            self._future_prediction(gx, sample, present, dec_in, B)
            n_dec_t = self.nf + 1
        else:
            dec_in = present
            n_dec_t = 1
        out.update(self._decoder(dec_in, B, n_dec_t, into))
        return out

    def _future_prediction(self, gx, sample, present, dec_in, B):
        lib = self.lib
        nf, H, W, ch = self.nf, present.H, present.W, self.state_c
        g0 = self.grus[0]
        gates_bias = tilde_bias = None
        if g0.const_x:
            # nine bias rows per sample: what the constant latent channels contribute for each border class
            gates_bias = self.vec('gru0_gates_bias', B, 9 * 2 * ch)
            tilde_bias = self.vec('gru0_tilde_bias', B, 9 * ch)
            lib.rowwise_dense(sample, self.latent, B, self.latent, g0.gates_xw, self.latent, 0, 9 * 2 * ch, None, None, NONE,
                              False, gates_bias, 9 * 2 * ch)
            lib.rowwise_dense(sample, self.latent, B, self.latent, g0.tilde_xw, self.latent, 0, 9 * ch, None, None, NONE,
                              False, tilde_bias, 9 * ch)
        seq_in = None
        n_blocks = len(self.grus)
        for i, gru in enumerate(self.grus):
            O = self.buf(f'gru{i}_O', B * nf, H, W, ch)
            U = self.buf('gru_U', B, H, W, ch)
            RH = self.buf('gru_RH', B, H, W, ch)
            for t in range(nf):
                h_t = present if t == 0 else O.images(t - 1, B, step=nf)
                o_t = O.images(t, B, step=nf)
                if i == 0 and gru.const_x:
                    gru.gates([h_t], U, out2=RH, aux0=h_t, img_bias=gates_bias, img_bias_border=True)
                    gru.tilde([RH], o_t, aux0=U, aux1=h_t, img_bias=tilde_bias, img_bias_border=True)
                    continue
                x_t = gx if i == 0 else seq_in.images(t, B, step=nf)
                gru.gates([x_t, h_t], U, out2=RH, aux0=h_t)
                gru.tilde([x_t, RH], o_t, aux0=U, aux1=h_t)
            x = O
            blocks = self.res_blocks[i]
            t1_ready = None
            for k, blk in enumerate(blocks):
                last = (i == n_blocks - 1) and (k == len(blocks) - 1)
                # a block whose tail also computes the next block's down-projection hands it over in `next_t1`
                next_t1 = None
                if k + 1 < len(blocks) and blk.next_tail is not None:
                    next_t1 = self.buf(f'fpbn_next{k % 2}', B * nf, H, W, blocks[k + 1].mid)
                if not last:
                    y = self.buf(f'res_{(k % 2)}', B * nf, H, W, ch)
                    blk.run(self, [x], y, self._scratch('fp'), t1_ready=t1_ready, next_t1=next_t1)
                    x = y
                    t1_ready = next_t1
                else:
                    # the last convolution writes straight into frames 1.. of the decoder input, per batch element
                    t1 = t1_ready
                    if t1 is None:
                        t1 = self.buf('fpbn_t1', B * nf, H, W, blk.mid)
                        blk.conv1([x], t1)
                    if blk.fused_tail is None:
                        t2 = self.buf('fpbn_t2', B * nf, H, W, blk.mid)
                        blk.conv2([t1], t2)
                    for b in range(B):
                        dst, res = dec_in.images(b * (nf + 1) + 1, nf), x.images(b * nf, nf)
                        if blk.fused_tail is None:
                            blk.conv3([t2.images(b * nf, nf)], dst, res=res)
                        else:
                            blk.fused_tail([t1.images(b * nf, nf)], dst, res=res)
            seq_in = x
        if not self.res_blocks[-1]:
            dec_in.nhwc().view(B, nf + 1, H, W, -1)[:, 1:].copy_(seq_in.nhwc().view(B, nf, H, W, -1))

    def _decoder(self, x, B, T, into=None):
        into = into or {}
        lib = self.lib
        n, H, W = x.n_img, x.H, x.W
        skips = [x]
        h1, w1 = self.dec_first.out_hw(H, W)
        y = self.buf('dec_stem', n, h1, w1, 64)
        self.dec_first([x], y)
        for li, blocks in enumerate(self.dec_layers):
            for bi, ops in enumerate(blocks):
                ho, wo = ops['conv1'].out_hw(y.H, y.W)
                t = self.buf(f'dec_l{li}_t', n, ho, wo, ops['cout'])
                ops['conv1']([y], t)
                ident = y
                if ops['down'] is not None:
                    ident = self.buf(f'dec_l{li}_ds', n, ho, wo, ops['cout'])
                    ops['down']([y], ident)
                o = self.buf(f'dec_l{li}_o{bi}', n, ho, wo, ops['cout'])
                ops['conv2']([t], o, res=ident)
                y = o
            if li < 2:
                skips.append(y)
        for ui, up in enumerate(self.dec_ups):
            skip = skips[2 - ui]
            low = self.buf(f'dec_up{ui}_low', n, y.H, y.W, up['cout'])
            up['conv']([y], low)
            o = self.buf(f'dec_up{ui}', n, skip.H, skip.W, up['cout'])
            lib.upsample2x_add(low, low.ld, n, low.H, low.W, up['cout'], up['shift'], skip, skip.ld, o, o.ld)
            y = o
        out = {}
        if self.heads_fused:
            results = [into[hd['name']].view(n, hd['n_out'], H, W) if hd['name'] in into else
                       torch.empty(n, hd['n_out'], H, W, dtype=torch.float32, device=self.device) for hd in self.heads_final]
            planes = [(res.data_ptr() + 4 * j * H * W, hd['n_out'] * H * W)
                      for res, hd in zip(results, self.heads_final) for j in range(hd['n_out'])]
            self.heads_conv([y], HeadsOut(n, H, W, results[0]), head_planes=planes)
            for res, hd in zip(results, self.heads_final):
                out[hd['name']] = res.view(B, T, hd['n_out'], H, W)
        else:
            hb = self.buf('dec_heads', n, H, W, self.heads_conv.cout)
            self.heads_conv([y], hb)
            for hd in self.heads_final:
                res = (into[hd['name']].view(n, hd['n_out'], H, W) if hd['name'] in into else
                       torch.empty(n, hd['n_out'], H, W, dtype=torch.float32, device=self.device))
                lib.heads_1x1_nchw(hb.slice(hd['c_off'], self.head_c), hb.ld, n, H * W, self.head_c, self.head_c,
                                   hd['w'], hd['b'], [0] * hd['n_out'], [hd['sigmoid']] * hd['n_out'], res)
                out[hd['name']] = res.view(B, T, hd['n_out'], H, W)
        if 'instance_flow' not in out:
            out['instance_flow'] = None
        return out
