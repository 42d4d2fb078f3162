"""The implicit-GEMM convolution *source* (fiery_amd/csrc/conv_igemm.hip) on the CPU simulator versus
torch's fp32 convolutions: MFMA fragment mapping, LDS swizzle, im2col gather (strides, padding, causal
time taps, two-source concat), weight packing and every epilogue, without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fiery_amd import native
from fiery_amd.ops import Buf, ConvOp, HeadsOut, identity_chan_map, round_up

TOL = dict(rtol=2e-5, atol=2e-5)


def _to_buf(x_nchw, width=None):
    n, c, h, w = x_nchw.shape
    width = width or round_up(c, 8)
    t = torch.zeros(n, h, w, width)
    t[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return Buf(t, n, h, w, width)


@pytest.mark.parametrize('cin,cout,k,stride,hw', [
    (16, 64, 3, 1, (9, 14)),      # BN=64 tile, 3x3
    (13, 32, 3, 2, (11, 9)),      # ragged channels, BN=32, stride 2, odd size
    (24, 70, 1, 1, (7, 20)),      # 1x1, cout padded to 96 (three BN=32 tiles)
    (8, 40, 7, 2, (12, 12)),      # 7x7 stride 2 like the decoder stem, cout padded to 64
    (24, 128, 3, 1, (10, 13)),    # BN=128 tile: 64x64 per wavefront
    (16, 250, 1, 1, (6, 22)),     # cout padded to 256: two BN=128 tiles
])
def test_conv2d_bn_relu(sim, cin, cout, k, stride, hw):
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.2
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    src = _to_buf(x)
    op = ConvOp(sim, w, identity_chan_map(cin), (src.C // 8, 0), scale, shift, 'cpu', stride=stride, act=native.ACT_RELU)
    ho, wo = op.out_hw(*hw)
    out = Buf.alloc(2, ho, wo, cout, 'cpu')
    op([src], out)
    want = F.relu(F.conv2d(x, w, stride=stride, padding=(k - 1) // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    got = out.to_nchw()[:, :cout]
    assert torch.allclose(got, want, **TOL), (got - want).abs().max()
    # padded output channels are written as exact zeros (they feed later convs as inputs)
    assert out.to_nchw()[:, cout:].abs().max() == 0 if out.C > cout else True


@pytest.mark.parametrize('tile_m', ['64', '128'])
@pytest.mark.parametrize('cout', [64, 128])
def test_both_tile_heights(sim, monkeypatch, tile_m, cout):
    """The 64- and 128-pixel tile variants are chosen by a fill heuristic; force each and compare."""
    monkeypatch.setenv('FIERY_CONV_TILE_M', tile_m)
    g = torch.Generator().manual_seed(int(tile_m) + cout)
    x = torch.randn(1, 16, 11, 19, generator=g)               # 209 pixels: ragged last tile either way
    w = torch.randn(cout, 16, 3, 3, generator=g) * 0.2
    src = _to_buf(x)
    op = ConvOp(sim, w, identity_chan_map(16), (2, 0), torch.ones(cout), torch.zeros(cout), 'cpu', act=native.ACT_RELU)
    out = Buf.alloc(1, 11, 19, cout, 'cpu')
    op([src], out)
    assert torch.allclose(out.to_nchw(), F.relu(F.conv2d(x, w, padding=1)), **TOL)


def test_asymmetric_weights_catch_transposes(sim):
    """A = identity-like input with an asymmetric weight matrix: a row/column swap in the MFMA output
    mapping or the packing cannot pass."""
    cin, cout = 32, 64
    x = torch.zeros(1, cin, 4, 8)
    for c in range(cin):
        x[0, c, c % 4, c % 8] = 1.0 + c
    w = torch.arange(cout * cin, dtype=torch.float32).view(cout, cin, 1, 1) / 100.0
    src = _to_buf(x)
    op = ConvOp(sim, w, identity_chan_map(cin), (4, 0), torch.ones(cout), torch.zeros(cout), 'cpu')
    out = Buf.alloc(1, 4, 8, cout, 'cpu')
    op([src], out)
    assert torch.allclose(out.to_nchw(), F.conv2d(x, w), **TOL)


def test_two_source_concat_with_residual_and_bias(sim):
    g = torch.Generator().manual_seed(5)
    xa = torch.randn(2, 12, 6, 10, generator=g)      # source 0: 12 real channels in 16
    xb = torch.randn(2, 20, 6, 10, generator=g)      # source 1: 20 real channels in 24
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    res = torch.randn(2, 64, 6, 10, generator=g)
    img_bias = torch.randn(2, 64, generator=g)
    a, b, r = _to_buf(xa), _to_buf(xb), _to_buf(res)
    cmap = identity_chan_map(12) + identity_chan_map(20, offset=16)
    for res_pre in (False, True):
        op = ConvOp(sim, w, cmap, (2, 3), torch.ones(64), torch.zeros(64), 'cpu', act=native.ACT_RELU,
                    res_before_act=res_pre)
        out = Buf.alloc(2, 6, 10, 64, 'cpu')
        op([a, b], out, res=r, img_bias=img_bias)
        y = F.conv2d(torch.cat([xa, xb], 1), w, padding=1) + img_bias.view(2, 64, 1, 1)
        want = F.relu(y + res) if res_pre else F.relu(y) + res
        assert torch.allclose(out.to_nchw(), want, **TOL)


def border_class_bias(w_const, vec):
    """[9][cout] table: contribution of spatially constant input channels `vec` (cin_c,) through the 3x3 weights
    `w_const` (cout, cin_c, 3, 3) for the nine border classes of a zero-padded, stride-1 convolution."""
    rows = []
    for cy in range(3):
        kys = [1, 2] if cy == 0 else [0, 1] if cy == 2 else [0, 1, 2]
        for cx in range(3):
            kxs = [1, 2] if cx == 0 else [0, 1] if cx == 2 else [0, 1, 2]
            wsum = w_const[:, :, kys][:, :, :, kxs].sum(dim=(2, 3))           # (cout, cin_c)
            rows.append(wsum @ vec)
    return torch.stack(rows)


@pytest.mark.parametrize('unaligned', [False, True])
def test_constant_channels_folded_into_border_class_bias(sim, monkeypatch, unaligned):
    """cat[const, h] through a 3x3 convolution == h-part convolution + one of nine per-image bias rows."""
    if unaligned:
        monkeypatch.setenv('FIERY_CONV_VEC_EPILOGUE', '0')         # register epilogue path
    g = torch.Generator().manual_seed(77)
    n, cc, ch, cout, hw = 2, 8, 16, 64, (7, 9)
    vec = torch.randn(n, cc, generator=g)
    h = torch.randn(n, ch, *hw, generator=g)
    w = torch.randn(cout, cc + ch, 3, 3, generator=g) * 0.2
    shift = torch.randn(cout, generator=g)
    full = torch.cat([vec.view(n, cc, 1, 1).expand(n, cc, *hw), h], 1)
    want = torch.sigmoid(F.conv2d(full, w, padding=1) + shift.view(1, -1, 1, 1))
    op = ConvOp(sim, w[:, cc:], identity_chan_map(ch), (ch // 8, 0), torch.ones(cout), shift, 'cpu', act=native.ACT_SIGMOID)
    table = torch.stack([border_class_bias(w[:, :cc], vec[i]) for i in range(n)]).contiguous()      # (n, 9, cout)
    out = Buf.alloc(n, *hw, cout, 'cpu')
    op([_to_buf(h)], out, img_bias=table, img_bias_border=True)
    assert torch.allclose(out.to_nchw(), want, **TOL), (out.to_nchw() - want).abs().max()


@pytest.mark.parametrize('form', [None, 'wino', 'wino8'])
def test_decoder_heads_epilogue_keeps_the_hidden_tensor_on_chip(sim, monkeypatch, form):
    """Four heads (Conv3x3 -> BN -> ReLU -> Conv1x1 + bias [-> Sigmoid], models/decoder.py:30-51) as one 256-cout GEMM
    whose epilogue stores only the seven final rows, as NCHW planes of four separate tensors.  form = 'wino': the same through
    the Winograd F(2x2, 3x3) kernel, whose 64-cout workgroup tile is one head's hidden channels (round 5)."""
    g = torch.Generator().manual_seed(5)
    n, cin, hw = 2, 16, (9, 11)                       # 99 pixels per image: tiles straddle the image boundary
    n_outs, sig = [2, 1, 2, 2], [False, True, False, False]
    x = torch.randn(n, cin, *hw, generator=g)
    w3 = torch.randn(4 * 64, cin, 3, 3, generator=g) * 0.15
    scale = torch.rand(4 * 64, generator=g) + 0.5
    shift = torch.randn(4 * 64, generator=g) * 0.3
    w1 = [torch.randn(k, 64, generator=g) * 0.2 for k in n_outs]
    b1 = [torch.randn(k, generator=g) for k in n_outs]
    op = ConvOp(sim, w3, identity_chan_map(cin), (cin // 8, 0), scale, shift, 'cpu', act=native.ACT_RELU)
    groups = [h for h, k in enumerate(n_outs) for _ in range(k)]
    op.attach_heads(torch.cat(w1), torch.cat(b1), groups, [sig[h] for h in groups])
    monkeypatch.setenv('FIERY_WINOGRAD_WAVES', '8' if form == 'wino8' else '4')
    op.force_form = 'wino' if form == 'wino8' else form
    tol = TOL if form is None else dict(rtol=2e-5, atol=2e-5)
    outs = [torch.full((n, k, *hw), float('nan')) for k in n_outs]
    hwp = hw[0] * hw[1]
    planes = [(outs[h].data_ptr() + 4 * j * hwp, n_outs[h] * hwp) for h in range(4) for j in range(n_outs[h])]
    op([_to_buf(x)], HeadsOut(n, *hw, outs[0]), head_planes=planes)
    hidden = F.relu(F.conv2d(x, w3, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    for h in range(4):
        want = F.conv2d(hidden[:, 64 * h:64 * h + 64], w1[h].view(-1, 64, 1, 1), b1[h])
        if sig[h]:
            want = torch.sigmoid(want)
        assert torch.allclose(outs[h], want, **tol), (h, (outs[h] - want).abs().max())


def test_channel_slices_of_wider_buffers(sim):
    """Reads a channel slice of a wide buffer and writes into a slice of another (concat by placement)."""
    g = torch.Generator().manual_seed(6)
    wide = torch.randn(1, 5, 7, 48, generator=g)
    src = Buf(wide, 1, 5, 7, 48).slice(16, 16)
    w = torch.randn(32, 16, 1, 1, generator=g)
    dst_t = torch.full((1, 5, 7, 96), 7.0)
    dst = Buf(dst_t, 1, 5, 7, 96).slice(32, 32)
    op = ConvOp(sim, w, identity_chan_map(16), (2, 0), torch.ones(32), torch.zeros(32), 'cpu')
    op([src], dst)
    want = F.conv2d(wide[..., 16:32].permute(0, 3, 1, 2), w)
    assert torch.allclose(dst_t[..., 32:64].permute(0, 3, 1, 2), want, **TOL)
    assert (dst_t[..., :32] == 7.0).all() and (dst_t[..., 64:] == 7.0).all()


@pytest.mark.parametrize('kt', [2, 1])
def test_causal_conv3d_with_time_window(sim, kt):
    """(kT,3,3) causal convolution computing only output frames t >= t_out0 (dead-frame pruning)."""
    g = torch.Generator().manual_seed(8 + kt)
    B, T, cin, cout, H, W = 2, 3, 16, 32, 5, 6
    x = torch.randn(B, cin, T, H, W, generator=g)
    w = torch.randn(cout, cin, kt, 3, 3, generator=g) * 0.2
    want = F.conv3d(F.pad(x, (1, 1, 1, 1, kt - 1, 0)), w)                  # (B, cout, T, H, W)
    x_img = x.permute(0, 2, 1, 3, 4).reshape(B * T, cin, H, W)               # images ordered (b, t)
    src = _to_buf(x_img)
    op = ConvOp(sim, w, identity_chan_map(cin), (2, 0), torch.ones(cout), torch.zeros(cout), 'cpu')
    for t_out0 in (0, 1):
        T_out = T - t_out0
        out = Buf.alloc(B * T_out, H, W, cout, 'cpu')
        op([(src, T * src.img_stride, src.img_stride)], out, T_out=T_out, t_out0=t_out0, t_in_add=t_out0)
        got = out.to_nchw().view(B, T_out, cout, H, W).permute(0, 2, 1, 3, 4)
        assert torch.allclose(got, want[:, :, t_out0:], **TOL)


def test_gru_epilogues(sim):
    """conv_update|conv_reset fused into one N=128 GEMM, then the state_tilde conv with the gate
    arithmetic in its epilogue: one full SpatialGRU cell (fiery/layers/temporal.py:49-62)."""
    g = torch.Generator().manual_seed(11)
    B, cx, ch, H, W = 2, 32, 64, 6, 9
    x = torch.randn(B, cx, H, W, generator=g)
    h = torch.randn(B, ch, H, W, generator=g)
    wu = torch.randn(ch, cx + ch, 3, 3, generator=g) * 0.05
    wr = torch.randn(ch, cx + ch, 3, 3, generator=g) * 0.05
    bu, br = torch.randn(ch, generator=g), torch.randn(ch, generator=g)
    wt = torch.randn(ch, cx + ch, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(ch, generator=g) + 0.5, torch.randn(ch, generator=g)
    xs = torch.cat([x, h], 1)
    u = torch.sigmoid(F.conv2d(xs, wu, bu, padding=1))
    r = torch.sigmoid(F.conv2d(xs, wr, br, padding=1))
    tilde = F.relu(F.conv2d(torch.cat([x, (1 - r) * h], 1), wt, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    want = (1 - u) * h + u * tilde

    xb, hb = _to_buf(x), _to_buf(h)
    cmap = identity_chan_map(cx) + identity_chan_map(ch, offset=cx)
    gates = ConvOp(sim, torch.cat([wu, wr], 0), cmap, (cx // 8, ch // 8), torch.ones(2 * ch), torch.cat([bu, br]),
                   'cpu', epi=native.EPI_GRU_GATES)
    ubuf, rh = Buf.alloc(B, H, W, ch, 'cpu'), Buf.alloc(B, H, W, ch, 'cpu')
    gates([xb, hb], ubuf, out2=rh, aux0=hb)
    assert torch.allclose(ubuf.to_nchw(), u, **TOL)
    assert torch.allclose(rh.to_nchw(), (1 - r) * h, **TOL)
    tilde_op = ConvOp(sim, wt, cmap, (cx // 8, ch // 8), sc, sh, 'cpu', act=native.ACT_RELU, epi=native.EPI_GRU_OUT)
    out, out2 = Buf.alloc(B, H, W, ch, 'cpu'), Buf.alloc(B, H, W, ch, 'cpu')
    tilde_op([xb, rh], out, out2=out2, aux0=ubuf, aux1=hb)
    assert torch.allclose(out.to_nchw(), want, **TOL)
    assert torch.equal(out.to_nchw(), out2.to_nchw())


def test_strided_image_views_for_time_steps(sim):
    """Reading time step t of a (batch, time) sequence buffer and writing step t of another."""
    g = torch.Generator().manual_seed(12)
    B, T, c, H, W = 2, 3, 16, 4, 5
    seq = torch.randn(B * T, H, W, c, generator=g)
    sbuf = Buf(seq, B * T, H, W, c)
    w = torch.randn(32, c, 3, 3, generator=g) * 0.2
    op = ConvOp(sim, w, identity_chan_map(c), (2, 0), torch.ones(32), torch.zeros(32), 'cpu')
    dst = Buf.alloc(B * T, H, W, 32, 'cpu')
    t = 1
    op([sbuf.images(t, B, step=T)], dst.images(t, B, step=T))
    x_t = seq.view(B, T, H, W, c)[:, t].permute(0, 3, 1, 2)
    got = dst.nhwc().view(B, T, H, W, 32)[:, t].permute(0, 3, 1, 2)
    assert torch.allclose(got, F.conv2d(x_t, w, padding=1), **TOL)
    assert dst.nhwc().view(B, T, H, W, 32)[:, 0].abs().max() == 0


def test_heads_1x1_nchw(sim):
    g = torch.Generator().manual_seed(13)
    n, hw, C, head_c = 2, 70, 64, 16
    x = torch.randn(n, hw, C, generator=g)
    c_off = [0, 0, 16, 32, 32, 48, 48]
    sig = [0, 0, 1, 0, 0, 0, 0]
    w = torch.randn(7, head_c, generator=g)
    b = torch.randn(7, generator=g)
    out = torch.empty(n, 7, hw)
    sim.heads_1x1_nchw(x, C, n, hw, C, head_c, w, b, c_off, sig, out)
    for o in range(7):
        want = (x[:, :, c_off[o]:c_off[o] + head_c] * w[o]).sum(-1) + b[o]
        if sig[o]:
            want = torch.sigmoid(want)
        assert torch.allclose(out[:, o], want, **TOL)


@pytest.mark.parametrize('form', [None, 'split'])
@pytest.mark.parametrize('epilogue', ['rows', 'rows, images apart', 'unaligned'])
@pytest.mark.parametrize('mid,cout,stride', [(32, 64, 1), (20, 40, 2)])
def test_chained_pointwise_conv_with_residual(sim, monkeypatch, mid, cout, stride, epilogue, form):
    """3x3 conv + BN + ReLU -> 1x1 conv + BN + ReLU + residual as ONE kernel (the Bottleneck's tail,
    fiery/layers/convolutions.py:123-168): the intermediate values stay on chip (round 5: in the accumulator registers).  The
    three ways its rows leave: full 16-byte rows of dense tensors, of tensors whose images are apart (per-row image / pixel
    split), and channel by channel from the registers when the destinations are not 16-byte addressable.  form = 'split'
    (round 6): the 3 x 3 product on the bf16 matrix cores with three-term operands (the 20-channel case has no whole stages and
    stays on the fp32 kernel)."""
    if form == 'split' and mid != 32:
        pytest.skip('the split form needs whole 32-channel stages')
    if epilogue == 'unaligned':
        monkeypatch.setenv('FIERY_CONV_VEC_EPILOGUE', '0')
    elif epilogue != 'rows':
        monkeypatch.setenv('FIERY_CONV_DENSE_EPILOGUE', '0')
    g = torch.Generator().manual_seed(mid + cout)
    x = torch.randn(2, mid, 9, 14, generator=g)
    w2 = torch.randn(mid, mid, 3, 3, generator=g) * 0.15
    w3 = torch.randn(cout, mid, 1, 1, generator=g) * 0.3
    s2, b2 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g)
    s3, b3 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    src = _to_buf(x)
    op = ConvOp(sim, w2, identity_chan_map(mid), (src.C // 8, 0), s2, b2, 'cpu', stride=stride, act=native.ACT_RELU)
    op.chain_pointwise(w3, s3, b3, native.ACT_RELU)
    op.force_form = form
    ho, wo = op.out_hw(9, 14)
    res = torch.randn(2, cout, ho, wo, generator=g)
    out = Buf.alloc(2, ho, wo, cout, 'cpu')
    op([src], out, res=_to_buf(res))
    assert op.last_form == (form or 0)
    h = F.relu(F.conv2d(x, w2, stride=stride, padding=1) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1))
    want = F.relu(F.conv2d(h, w3) * s3.view(1, -1, 1, 1) + b3.view(1, -1, 1, 1)) + res
    assert torch.allclose(out.to_nchw()[:, :cout], want, **TOL), (out.to_nchw()[:, :cout] - want).abs().max()


@pytest.mark.parametrize('form', [None, 'split'])
@pytest.mark.parametrize('shape', [(2, 9, 14), (1, 16, 16), (3, 5, 43)])
def test_chained_tail_plus_the_next_blocks_down_projection(sim, shape, form):
    """Bottleneck tail (3x3 + 1x1 + residual) AND the following Bottleneck's 1x1 down-projection (64 -> 32, BN, ReLU) in
    one kernel: both outputs against torch; image counts and sizes that make tiles straddle images and end ragged."""
    n, H, W = shape
    g = torch.Generator().manual_seed(n * 100 + H)
    mid, cout = 32, 64
    x = torch.randn(n, mid, H, W, generator=g)
    w2 = torch.randn(mid, mid, 3, 3, generator=g) * 0.15
    w3 = torch.randn(cout, mid, 1, 1, generator=g) * 0.3
    w4 = torch.randn(mid, cout, 1, 1, generator=g) * 0.2
    s2, b2 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g)
    s3, b3 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    s4, b4 = torch.rand(mid, generator=g) + 0.5, torch.randn(mid, generator=g)
    src = _to_buf(x)
    base = ConvOp(sim, w2, identity_chan_map(mid), (src.C // 8, 0), s2, b2, 'cpu', act=native.ACT_RELU)
    base.chain_pointwise(w3, s3, b3, native.ACT_RELU)
    op = base.chain_next(w4, s4, b4, native.ACT_RELU)
    op.force_form = base.force_form = form
    res = torch.randn(n, cout, H, W, generator=g)
    out = Buf.alloc(n, H, W, cout, 'cpu')
    nxt = Buf.alloc(n, H, W, mid, 'cpu')
    op([src], out, res=_to_buf(res), out3=nxt)
    assert op.last_form == (form or 0)
    h = F.relu(F.conv2d(x, w2, padding=1) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1))
    y = F.relu(F.conv2d(h, w3) * s3.view(1, -1, 1, 1) + b3.view(1, -1, 1, 1)) + res
    t = F.relu(F.conv2d(y, w4) * s4.view(1, -1, 1, 1) + b4.view(1, -1, 1, 1))
    assert torch.allclose(out.to_nchw()[:, :cout], y, **TOL), (out.to_nchw()[:, :cout] - y).abs().max()
    assert torch.allclose(nxt.to_nchw()[:, :mid], t, **TOL), (nxt.to_nchw()[:, :mid] - t).abs().max()
    # the plain op still works on its own
    out_b = Buf.alloc(n, H, W, cout, 'cpu')
    base([src], out_b, res=_to_buf(res))
    assert torch.equal(out_b.nhwc(), out.nhwc())


@pytest.mark.parametrize('case', ['64->64 3x3', '64->64 7x7 stride 2', '32->32 3x3 ragged', '64+32->32 1x1 two sources', '64->40 3x3 residual',
                                  '32->32 (2,3,3) temporal'])
def test_split_tile_form_equals_torch(sim, case):
    """The split tile kernels (round 6, FIERY_PRECISION_F32_SPLIT: every operand as three bf16 terms, six partial products per
    product on the bf16 matrix cores - an fp32-accurate form, held to the fp32 kernels' tolerance) against torch: both cout tile
    widths, strides, two sources, ragged tiles, a residual, a temporal kernel."""
    g = torch.Generator().manual_seed(len(case))
    if case == '32->32 (2,3,3) temporal':
        B, T, H, W, C = 2, 3, 6, 9, 32
        seq = torch.randn(B, T, C, H, W, generator=g)
        w = torch.randn(32, C, 2, 3, 3, generator=g) * 0.1
        op = ConvOp(sim, w, identity_chan_map(C), (C // 8, 0), torch.ones(32), torch.zeros(32), 'cpu', act=native.ACT_RELU)
        op.force_form = 'split'
        src = Buf(seq.permute(0, 1, 3, 4, 2).reshape(B * T, H, W, C).contiguous(), B * T, H, W, C)
        out = Buf.alloc(B * T, H, W, 32, 'cpu')
        op([(src, T * src.img_stride, src.img_stride)], out, T_out=T)
        assert op.last_form == 'split'
        xp = F.pad(seq.permute(0, 2, 1, 3, 4), (1, 1, 1, 1, 1, 0))                      # causal in time, 'same' in space
        want = F.relu(F.conv3d(xp, w)).permute(0, 2, 1, 3, 4).reshape(B * T, 32, H, W)
        assert torch.allclose(out.to_nchw(), want, **TOL), (out.to_nchw() - want).abs().max()
        return
    cfg = {'64->64 3x3': (3, 64, 0, 64, 3, 1, 11, 13), '64->64 7x7 stride 2': (2, 64, 0, 64, 7, 2, 20, 18), '32->32 3x3 ragged': (3, 32, 0, 32, 3, 1, 5, 43),
           '64+32->32 1x1 two sources': (2, 64, 32, 32, 1, 1, 9, 14), '64->40 3x3 residual': (2, 64, 0, 40, 3, 1, 10, 12)}[case]
    n, c0, c1, cout, k, stride, H, W = cfg
    x0 = torch.randn(n, c0, H, W, generator=g)
    x1 = torch.randn(n, c1, H, W, generator=g) if c1 else None
    w = torch.randn(cout, c0 + c1, k, k, generator=g) / ((c0 + c1) * k * k) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    cmap = identity_chan_map(c0) + (identity_chan_map(c1, offset=c0) if c1 else [])
    op = ConvOp(sim, w, cmap, (c0 // 8, c1 // 8), sc, sh, 'cpu', stride=stride, act=native.ACT_RELU)
    op.force_form = 'split'
    ho, wo = op.out_hw(H, W)
    res = torch.randn(n, cout, ho, wo, generator=g) if 'residual' in case else None
    out = Buf.alloc(n, ho, wo, round_up(cout, 32) if 'round_up' in globals() else (cout + 31) // 32 * 32, 'cpu')
    op([_to_buf(x0)] + ([_to_buf(x1)] if c1 else []), out, res=_to_buf(res) if res is not None else None)
    assert op.last_form == 'split' and op.packed_split is not None
    x = torch.cat([x0, x1], 1) if c1 else x0
    want = F.relu(F.conv2d(x, w, stride=stride, padding=(k - 1) // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    if res is not None:
        want = want + res
    assert torch.allclose(out.to_nchw()[:, :cout], want, **TOL), (out.to_nchw()[:, :cout] - want).abs().max()


@pytest.mark.parametrize('case', range(20))
def test_conv_random_shapes(sim, case):
    """Seeded sweep: channel counts that are and are not whole 32-channel stages (scalar-addressed vs per-lane loop,
    small-cin loop), one or two sources, kernel 1 / 3 / 5, stride 1 / 2, ragged image sizes and counts, both tile
    heights, residual before / after the activation, all activations - against torch."""
    rng = np.random.RandomState(500 + case)
    c0 = int(rng.choice([8, 16, 24, 32, 35, 64, 96]))
    c1 = int(rng.choice([0, 0, 8, 32, 64]))
    cout = int(rng.choice([8, 24, 32, 40, 64, 100, 128]))
    k = int(rng.choice([1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 2]))
    n, H, W = int(rng.randint(1, 4)), int(rng.randint(3, 14)), int(rng.randint(3, 20))
    act = int(rng.choice([native.ACT_NONE, native.ACT_RELU, native.ACT_SIGMOID, native.ACT_SWISH]))
    g = torch.Generator().manual_seed(case)
    x0 = torch.randn(n, c0, H, W, generator=g)
    x1 = torch.randn(n, c1, H, W, generator=g) if c1 else None
    cin = c0 + c1
    w = torch.randn(cout, cin, k, k, generator=g) * (0.5 / (cin ** 0.5 * k))
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    s0 = _to_buf(x0)
    srcs, units = [s0], (s0.C // 8, 0)
    cmap = identity_chan_map(c0)
    if c1:
        s1 = _to_buf(x1)
        srcs.append(s1)
        units = (s0.C // 8, s1.C // 8)
        cmap = cmap + identity_chan_map(c1, offset=s0.C)
    res_before = bool(rng.randint(2)) and act != native.ACT_SWISH
    op = ConvOp(sim, w, cmap, units, scale, shift, 'cpu', stride=stride, act=act, res_before_act=res_before)
    ho, wo = op.out_hw(H, W)
    use_res = bool(rng.randint(2))
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    out = Buf.alloc(n, ho, wo, cout, 'cpu')
    import os
    os.environ['FIERY_CONV_TILE_M'] = str(int(rng.choice([64, 128])))
    try:
        op(srcs, out, res=_to_buf(res) if use_res else None)
    finally:
        del os.environ['FIERY_CONV_TILE_M']
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    y = F.conv2d(xin, w, stride=stride, padding=(k - 1) // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if use_res and res_before:
        y = y + res
    y = {native.ACT_NONE: lambda v: v, native.ACT_RELU: F.relu, native.ACT_SIGMOID: torch.sigmoid,
         native.ACT_SWISH: lambda v: v * torch.sigmoid(v)}[act](y)
    if use_res and not res_before:
        y = y + res
    got = out.to_nchw()[:, :cout]
    assert torch.allclose(got, y, **TOL), (case, c0, c1, cout, k, stride, n, H, W, act, (got - y).abs().max().item())


# ---- bf16 matrix-core form (v_mfma_f32_32x32x16_bf16; activations fp32 in memory, operands rounded on chip) ------------
def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('cin,cout,k,stride,hw,tile_m', [
    (32, 64, 3, 1, (9, 14), None),       # 128 x 64 or 64 x 64 tile
    (64, 32, 3, 1, (10, 12), None),      # BN = 32: half the threads fetch the 2 KiB of weights
    (32, 128, 1, 1, (7, 20), '64'),      # 64 x 128
    (64, 128, 3, 2, (11, 13), '128'),    # 128 x 128, stride 2, odd size
    (64, 64, 7, 2, (12, 12), '64'),      # 7 x 7 stride 2 like the decoder stem
])
def test_conv2d_bf16_form(sim, monkeypatch, cin, cout, k, stride, hw, tile_m):
    """The bf16 form against a convolution of the bf16-rounded operands in fp32 (exact products, fp32 sums: what the
    matrix core computes up to the order of additions), and - loosely - against the fp32 convolution it approximates."""
    if tile_m:
        monkeypatch.setenv('FIERY_CONV_TILE_M', tile_m)
    g = torch.Generator().manual_seed(cin * 10 + cout + k)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    src = _to_buf(x)
    op = ConvOp(sim, w, identity_chan_map(cin), (src.C // 8, 0), scale, shift, 'cpu', stride=stride, act=native.ACT_RELU,
                precision=native.PRECISION_BF16)
    ho, wo = op.out_hw(*hw)
    out = Buf.alloc(2, ho, wo, cout, 'cpu')
    op([src], out)
    pad = (k - 1) // 2
    want = F.relu(F.conv2d(_bf16(x), _bf16(w), stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    full = F.relu(F.conv2d(x, w, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    got = out.to_nchw()[:, :cout]
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-5), (got - want).abs().max()
    assert (got - full).abs().max() < 0.05 and (got - full).abs().max() > 1e-5       # it IS the rounded-operand result


@pytest.mark.parametrize('cin,cout,hw,n', [
    (32, 64, (9, 14), 2),        # one channel group, rows much shorter than a tile: a run covers several image rows
    (64, 128, (7, 5), 3),        # two groups, 64 x 128 tile, a tile spans two images (35 pixels each)
    (128, 64, (3, 70), 1),       # four groups, rows longer than a tile
    (96, 128, (1, 9), 2),        # one-row images: no tap above or below
    (32, 64, (66, 1), 1),        # one-column images: every left / right tap is outside
])
@pytest.mark.parametrize('tile_m', ['64', '128'])
def test_conv2d_bf16_halo_loop(sim, monkeypatch, cin, cout, hw, n, tile_m):
    """3 x 3 / stride 1 layers on 64-pixel tiles take the halo loop (the tile and its neighbours fetched once per 32-channel
    group, the nine taps as shifted windows of that block): against the rounded-operand convolution, and against the
    scalar-addressed loop it replaces (same products, another order of the fp32 additions)."""
    monkeypatch.setenv('FIERY_CONV_TILE_M', tile_m)
    g = torch.Generator().manual_seed(cin + cout + hw[0])
    x = torch.randn(n, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(n, cout, *hw, generator=g)
    src, rbuf = _to_buf(x), _to_buf(res)
    op = ConvOp(sim, w, identity_chan_map(cin), (src.C // 8, 0), scale, shift, 'cpu', act=native.ACT_RELU, precision=native.PRECISION_BF16)
    outs = {}
    for halo in ('1', '0'):
        monkeypatch.setenv('FIERY_CONV_HALO', halo)
        out = Buf.alloc(n, *hw, cout, 'cpu')
        op([src], out, res=rbuf)
        outs[halo] = out.to_nchw()[:, :cout]
    want = F.relu(F.conv2d(_bf16(x), _bf16(w), padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) + res
    assert torch.allclose(outs['1'], want, rtol=2e-5, atol=2e-5), (outs['1'] - want).abs().max()
    assert torch.allclose(outs['1'], outs['0'], rtol=1e-5, atol=1e-5)
    assert not torch.equal(outs['1'], outs['0']) or cin == 32        # (it really is the other loop: the sums round differently)


@pytest.mark.parametrize('cin,cout,hw,n', [(32, 64, (9, 14), 2), (64, 128, (7, 5), 3), (128, 64, (3, 70), 1), (96, 128, (1, 9), 2),
                                           (32, 64, (66, 1), 1)])
def test_conv2d_fp32_halo_loop(sim, monkeypatch, cin, cout, hw, n):
    """The halo loop in fp32 (FIERY_CONV_HALO_F32=1): the same windows, fp32 entries of 144 bytes, weights from global memory
    into each wavefront's MFMA operands - against the fp32 convolution and the scalar-addressed loop (another summation order)."""
    monkeypatch.setenv('FIERY_CONV_TILE_M', '64')
    g = torch.Generator().manual_seed(cin + cout + hw[0] + 1)
    x = torch.randn(n, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(n, cout, *hw, generator=g)
    src, rbuf = _to_buf(x), _to_buf(res)
    op = ConvOp(sim, w, identity_chan_map(cin), (src.C // 8, 0), scale, shift, 'cpu', act=native.ACT_RELU)
    outs = {}
    for halo in ('1', '0'):
        monkeypatch.setenv('FIERY_CONV_HALO_F32', halo)
        out = Buf.alloc(n, *hw, cout, 'cpu')
        op([src], out, res=rbuf)
        outs[halo] = out.to_nchw()[:, :cout]
    want = F.relu(F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) + res
    assert torch.allclose(outs['1'], want, **TOL), (outs['1'] - want).abs().max()
    assert torch.allclose(outs['1'], outs['0'], rtol=1e-5, atol=1e-5)
    assert not torch.equal(outs['1'], outs['0']) or cin == 32


def test_gru_epilogues_on_the_fp32_halo_loop(sim, monkeypatch):
    """Two sources (the GRU's [x, h]) and both GRU epilogues through the fp32 halo loop."""
    monkeypatch.setenv('FIERY_CONV_TILE_M', '64')
    monkeypatch.setenv('FIERY_CONV_HALO_F32', '1')
    g = torch.Generator().manual_seed(92)
    ch = 32
    x, h = torch.randn(2, ch, 8, 10, generator=g), torch.randn(2, ch, 8, 10, generator=g)
    wg = torch.randn(2 * ch, 2 * ch, 3, 3, generator=g) / (2 * ch * 9) ** 0.5
    bg = torch.randn(2 * ch, generator=g) * 0.1
    xb, hb = _to_buf(x), _to_buf(h)
    gates = ConvOp(sim, wg, identity_chan_map(ch) + identity_chan_map(ch, offset=ch), (ch // 8, ch // 8), torch.ones(2 * ch), bg,
                   'cpu', epi=native.EPI_GRU_GATES)
    U, RH = Buf.alloc(2, 8, 10, ch, 'cpu'), Buf.alloc(2, 8, 10, ch, 'cpu')
    gates([xb, hb], U, out2=RH, aux0=hb)
    pre = F.conv2d(torch.cat([x, h], 1), wg, padding=1) + bg.view(1, -1, 1, 1)
    assert torch.allclose(U.to_nchw(), torch.sigmoid(pre[:, :ch]), **TOL)
    assert torch.allclose(RH.to_nchw(), (1 - torch.sigmoid(pre[:, ch:])) * h, **TOL)


def test_bf16_form_falls_back_to_fp32_where_it_does_not_apply(sim):
    """13 input channels cannot take the scalar-addressed loop: the launch runs the fp32 kernel and is exact again."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 13, 9, 11, generator=g)
    w = torch.randn(32, 13, 3, 3, generator=g) * 0.2
    src = _to_buf(x)
    op = ConvOp(sim, w, identity_chan_map(13), (src.C // 8, 0), torch.ones(32), torch.zeros(32), 'cpu', precision=native.PRECISION_BF16)
    out = Buf.alloc(1, 9, 11, 32, 'cpu')
    op([src], out)
    assert torch.allclose(out.to_nchw()[:, :32], F.conv2d(x, w, padding=1), **TOL)


@pytest.mark.parametrize('tile_m', [None, '64', '128'])
def test_gru_epilogues_and_chained_tail_on_the_bf16_form(sim, monkeypatch, tile_m):
    """The epilogues are the fp32 kernel's: GRU gate GEMM (two sources; on 64-pixel tiles the halo loop, whose channel
    groups come from one source each) and a Bottleneck tail with its chained 1x1."""
    if tile_m:
        monkeypatch.setenv('FIERY_CONV_TILE_M', tile_m)
    g = torch.Generator().manual_seed(91)
    ch = 32
    x, h = torch.randn(1, ch, 8, 10, generator=g), torch.randn(1, ch, 8, 10, generator=g)
    wg = torch.randn(2 * ch, 2 * ch, 3, 3, generator=g) / (2 * ch * 9) ** 0.5
    bg = torch.randn(2 * ch, generator=g) * 0.1
    xb, hb = _to_buf(x), _to_buf(h)
    gates = ConvOp(sim, wg, identity_chan_map(ch) + identity_chan_map(ch, offset=ch), (ch // 8, ch // 8), torch.ones(2 * ch), bg,
                   'cpu', epi=native.EPI_GRU_GATES, precision=native.PRECISION_BF16)
    U, RH = Buf.alloc(1, 8, 10, ch, 'cpu'), Buf.alloc(1, 8, 10, ch, 'cpu')
    gates([xb, hb], U, out2=RH, aux0=hb)
    pre = F.conv2d(torch.cat([_bf16(x), _bf16(h)], 1), _bf16(wg), padding=1) + bg.view(1, -1, 1, 1)
    assert torch.allclose(U.to_nchw(), torch.sigmoid(pre[:, :ch]), rtol=2e-5, atol=2e-5)
    assert torch.allclose(RH.to_nchw(), (1 - torch.sigmoid(pre[:, ch:])) * h, rtol=2e-5, atol=2e-5)
    # Bottleneck tail: 3x3 32 -> 32 (+BN+ReLU) in bf16, chained 1x1 32 -> 64 (+BN+ReLU, + residual) on chip - since round 5 on
    # bf16 MFMAs too, its input taken from the accumulator registers and rounded there (fp32 accumulate, fp32 epilogue)
    t1 = torch.randn(1, 32, 8, 10, generator=g)
    w3 = torch.randn(32, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
    w1 = torch.randn(64, 32, 1, 1, generator=g) / 32 ** 0.5
    res = torch.randn(1, 64, 8, 10, generator=g)
    tail = ConvOp(sim, w3, identity_chan_map(32), (4, 0), torch.ones(32), torch.zeros(32), 'cpu', act=native.ACT_RELU,
                  precision=native.PRECISION_BF16).chain_pointwise(w1, torch.ones(64), torch.zeros(64), native.ACT_RELU)
    out = Buf.alloc(1, 8, 10, 64, 'cpu')
    tail([_to_buf(t1)], out, res=_to_buf(res))
    mid = F.relu(F.conv2d(_bf16(t1), _bf16(w3), padding=1))
    want = F.relu(F.conv2d(_bf16(mid), _bf16(w1))) + res
    assert torch.allclose(out.to_nchw(), want, rtol=3e-5, atol=3e-5), (out.to_nchw() - want).abs().max()
    # ... and with the next block's down-projection as a third stage (64 -> 32 on the finished values, residual included)
    w4 = torch.randn(32, 64, 1, 1, generator=g) / 8.0
    s4, b4 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    op3 = tail.chain_next(w4, s4, b4, native.ACT_RELU)
    out3, nxt = Buf.alloc(1, 8, 10, 64, 'cpu'), Buf.alloc(1, 8, 10, 32, 'cpu')
    op3([_to_buf(t1)], out3, res=_to_buf(res), out3=nxt)
    assert torch.allclose(out3.to_nchw(), want, rtol=3e-5, atol=3e-5)
    t = F.relu(F.conv2d(_bf16(out3.to_nchw()), _bf16(w4)) * s4.view(1, -1, 1, 1) + b4.view(1, -1, 1, 1))
    assert torch.allclose(nxt.to_nchw()[:, :32], t, rtol=5e-5, atol=5e-5), (nxt.to_nchw()[:, :32] - t).abs().max()


# ---- training: weight gradient ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,k,stride,hw', [(16, 32, 3, 1, (9, 14)), (24, 70, 1, 1, (6, 11)), (40, 64, 3, 2, (11, 9)),
                                                   (8, 130, 7, 2, (12, 12)),
                                                   # the LDS-staged 3 x 3 form: several segments, pixel-split wavefronts (32 x 32), a
                                                   # second channel tile with one sub-block, padded channels, one-row and odd maps
                                                   (32, 32, 3, 1, (9, 120)), (96, 64, 3, 1, (7, 60)), (35, 36, 3, 1, (6, 25)),
                                                   (128, 64, 3, 1, (5, 58)), (16, 32, 3, 1, (1, 9)), (64, 128, 3, 1, (13, 13)),
                                                   # the staged 1 x 1 form: flat 96-pixel chunks, ragged last chunk, narrow blocks, padded channels
                                                   (64, 32, 1, 1, (9, 14)), (32, 64, 1, 1, (7, 11)), (35, 36, 1, 1, (6, 25)), (128, 128, 1, 1, (5, 20)),
                                                   (70, 64, 1, 1, (1, 1))])
@pytest.mark.parametrize('precision', ['f32', 'split'])
def test_conv_wgrad_matches_autograd(sim, cin, cout, k, stride, hw, precision):
    """precision = 'split' (round 6, FIERY_PRECISION_F32_SPLIT): the staged 3 x 3 kernel's loop on the bf16 matrix cores with both
    operands as three bf16 terms - an fp32-accurate mode, same tolerance (kernels without the mode run fp32)."""
    if precision == 'split' and not (k == 3 and stride == 1):
        pytest.skip('the split mode belongs to the staged 3 x 3 kernel')
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g, requires_grad=True)
    pad = (k - 1) // 2
    y = F.conv2d(x, w, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    cin_pad = round_up(cin, 8)
    xb = torch.zeros(2, hw[0], hw[1], cin_pad)
    xb[..., :cin] = x.permute(0, 2, 3, 1)
    gb = gy.permute(0, 2, 3, 1).contiguous()
    dw = sim.conv_wgrad(xb, gb, cout, k, stride, pad, native.PRECISION_F32_SPLIT if precision == 'split' else native.PRECISION_F32)   # (cout, taps, cin_pad)
    got = dw[:, :, :cin].permute(0, 2, 1).reshape(cout, cin, k, k)
    assert torch.allclose(got, w.grad, rtol=1e-4, atol=1e-4), (got - w.grad).abs().max()
    assert dw[:, :, cin:].abs().max() == 0 if cin_pad > cin else True      # padded channels see zero inputs


@pytest.mark.parametrize('wgs', ['8', '16', '40'])
@pytest.mark.parametrize('case', ['3x3 64->128', '3x3 two sources -> 64, GRU out', '1x1 96->256 residual', '(2,3,3) 32->64 temporal'])
def test_stream_k_form_equals_the_tile_form(sim, monkeypatch, case, wgs):
    """Stream-K (`fiery_conv_desc.stream_k`, round 5): the launch's (tile, K chunk) units dealt evenly to a fixed number of
    workgroups, shared tiles summed through the workspace by the last arrival.  Against the tile form of the same launch and
    against torch: with 8 / 16 / 40 workgroups the same tiles are whole, split in two, and split in three or more parts; ragged
    pixel counts, several cout tiles, every epilogue kind the form covers.  (Visibility of the partials across XCDs is a GPU
    matter - tests/test_gpu_parity.py; here: indexing, part order, counters left at zero.)"""
    from fiery_amd import ops
    monkeypatch.setenv('FIERY_CONV_SK_WGS', wgs)
    g = torch.Generator().manual_seed(len(case) + int(wgs))
    if case == '3x3 64->128':
        x = torch.randn(2, 64, 13, 17, generator=g)
        w = torch.randn(128, 64, 3, 3, generator=g) / 24
        sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
        op = ConvOp(sim, w, identity_chan_map(64), (8, 0), sc, sh, 'cpu', act=native.ACT_RELU, tune=True)
        run = lambda o, out: o([_to_buf(x)], out)
        want = F.relu(F.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        shape = (2, 13, 17, 128)
    elif case == '3x3 two sources -> 64, GRU out':
        x, h = torch.randn(1, 32, 12, 15, generator=g), torch.randn(1, 64, 12, 15, generator=g)
        u = torch.rand(1, 64, 12, 15, generator=g)
        w = torch.randn(64, 96, 3, 3, generator=g) / 30
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        op = ConvOp(sim, w, identity_chan_map(32) + identity_chan_map(64, offset=32), (4, 8), sc, sh, 'cpu', act=native.ACT_RELU,
                    epi=native.EPI_GRU_OUT, tune=True)
        xb, hb, ub = _to_buf(x), _to_buf(h), _to_buf(u)
        run = lambda o, out: o([xb, hb], out, aux0=ub, aux1=hb)
        tilde = F.relu(F.conv2d(torch.cat([x, h], 1), w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        want = (1 - u) * h + u * tilde
        shape = (1, 12, 15, 64)
    elif case == '1x1 96->256 residual':
        x = torch.randn(3, 96, 9, 11, generator=g)
        w = torch.randn(256, 96, 1, 1, generator=g) / 10
        res = torch.randn(3, 256, 9, 11, generator=g)
        op = ConvOp(sim, w, identity_chan_map(96), (12, 0), torch.ones(256), torch.zeros(256), 'cpu', act=native.ACT_RELU, tune=True)
        rb = _to_buf(res)
        run = lambda o, out: o([_to_buf(x)], out, res=rb)
        want = F.relu(F.conv2d(x, w)) + res
        shape = (3, 9, 11, 256)
    else:
        B, T = 2, 3
        x = torch.randn(B * T, 32, 10, 9, generator=g)                       # frames of B sequences, t minor
        w = torch.randn(64, 32, 2, 3, 3, generator=g) / 24
        op = ConvOp(sim, w, identity_chan_map(32), (4, 0), torch.ones(64), torch.zeros(64), 'cpu', tune=True)
        xb = _to_buf(x)
        run = lambda o, out: o([(xb, T * xb.img_stride, xb.img_stride)], out, T_out=T)
        xs = x.view(B, T, 32, 10, 9).permute(0, 2, 1, 3, 4)
        want = F.conv3d(F.pad(xs, (1, 1, 1, 1, 1, 0)), w).permute(0, 2, 1, 3, 4).reshape(B * T, 64, 10, 9)
        shape = (B * T, 10, 9, 64)
    outs = {}
    for form in (128, 'sk'):
        op.force_form = form
        out = Buf.alloc(*shape, 'cpu')
        run(op, out)
        outs[form] = out.to_nchw()
    # (a case the plan does not cover with this many workgroups runs the tile form under 'sk' and never creates the workspace: which
    # worker process meets such a case first depends on how xdist deals the tests out)
    ws = ops._SK_WORKSPACES.get(('cpu', None, 0))
    if op.last_form == 'sk':
        assert ws is not None and int(ws['cnt'].abs().sum()) == 0, 'every launch leaves the ticket counters at zero'
    assert torch.allclose(outs['sk'], want, **TOL), (outs['sk'] - want).abs().max()
    assert torch.allclose(outs["sk"], outs[128], rtol=1e-5, atol=3e-6)        # (the same sums, split at other chunks)
    # a second launch through the same workspace (stale partials in every slot) gives the same bits
    out2 = Buf.alloc(*shape, 'cpu')
    run(op, out2)
    assert torch.equal(out2.to_nchw(), outs['sk'])


@pytest.mark.parametrize('epilogue', ['per kind', 'general', 'per kind, 8 wavefronts', 'per kind, split form', 'general, split form'])
@pytest.mark.parametrize('case', ['64->128 relu + residual, odd size', 'two sources -> gates', 'two sources -> GRU out', '32->64 border-class bias',
                                  '16->256 sequence views', '48->64 cout 40 stored'])
def test_winograd_form_equals_torch(sim, monkeypatch, case, epilogue):
    """Winograd F(2x2, 3x3) (`fiery_conv_desc.winograd`, csrc/conv_winograd.hip, round 5) against torch's direct convolution:
    odd image sizes (half-outside blocks), tiles that straddle images and end ragged, two sources, every epilogue kind the
    form covers, the border-class bias, several cout tiles, strided image views.  Tolerance: the transforms reorder the sums
    (fp32), so 2e-5 relative instead of the direct form's 1e-5.  epilogue = 'per kind': the kernels instantiated per epilogue
    kind (dense tensors: packed arithmetic, one buffer offset per tensor); 'general': the one kernel with everything behind
    run-time switches and per-pixel addressing that serves what those do not (here forced for every case); 'split form'
    (round 6, csrc/conv_winograd_split.hip): the K loop on the bf16 matrix cores with every operand as three bf16 terms - the
    same tolerance, it is an fp32-accurate form."""
    wform = 'wsplit' if 'split' in epilogue else 'wino'
    if 'general' in epilogue:
        monkeypatch.setenv('FIERY_WINOGRAD_GENERAL_EPILOGUE', '1')
    monkeypatch.setenv('FIERY_WINOGRAD_WAVES', '8' if '8 wavefronts' in epilogue else '4')
    g = torch.Generator().manual_seed(len(case))
    WTOL = dict(rtol=2e-5, atol=2e-5)

    def check(op, run, want, shape, cout):
        assert op.packed_winograd is not None and op.packed_winograd_split is not None
        outs = {}
        for form in (128, wform):
            op.force_form = form
            out = Buf.alloc(*shape, 'cpu')
            run(op, out)
            outs[form] = out.to_nchw()[:, :cout]
        assert torch.allclose(outs[128], want, **TOL)
        assert torch.allclose(outs[wform], want, **WTOL), (outs[wform] - want).abs().max()
        return outs

    if case == '64->128 relu + residual, odd size':
        x = torch.randn(3, 64, 13, 25, generator=g)
        w = torch.randn(128, 64, 3, 3, generator=g) / 24
        sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
        res = torch.randn(3, 128, 13, 25, generator=g)
        for pre in (False, True):
            op = ConvOp(sim, w, identity_chan_map(64), (8, 0), sc, sh, 'cpu', act=native.ACT_RELU, res_before_act=pre, tune=True)
            rb = _to_buf(res)
            y = F.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            want = F.relu(y + res) if pre else F.relu(y) + res
            check(op, lambda o, out: o([_to_buf(x)], out, res=rb), want, (3, 13, 25, 128), 128)
    elif case == 'two sources -> gates':
        ch = 32
        x, h = torch.randn(2, 64, 10, 12, generator=g), torch.randn(2, ch, 10, 12, generator=g)
        wg = torch.randn(2 * ch, 64 + ch, 3, 3, generator=g) / 30
        bg = torch.randn(2 * ch, generator=g) * 0.1
        op = ConvOp(sim, wg, identity_chan_map(64) + identity_chan_map(ch, offset=64), (8, ch // 8), torch.ones(2 * ch), bg, 'cpu',
                    epi=native.EPI_GRU_GATES, tune=True)
        xb, hb = _to_buf(x), _to_buf(h)
        pre = F.conv2d(torch.cat([x, h], 1), wg, padding=1) + bg.view(1, -1, 1, 1)
        op.force_form = wform
        U, RH = Buf.alloc(2, 10, 12, ch, 'cpu'), Buf.alloc(2, 10, 12, ch, 'cpu')
        op([xb, hb], U, out2=RH, aux0=hb)
        assert torch.allclose(U.to_nchw(), torch.sigmoid(pre[:, :ch]), **WTOL)
        assert torch.allclose(RH.to_nchw(), (1 - torch.sigmoid(pre[:, ch:])) * h, **WTOL)
        # ... and with per-image bias rows in border classes (the first SpatialGRU's folded constant input rides on the GATE
        # and OUTPUT epilogues too - round 5's first per-kind kernels forgot it there, and only the whole model noticed)
        bias = torch.randn(2, 9, 2 * ch, generator=g)
        cls = lambda v, size: torch.where(v == 0, 0, torch.where(v == size - 1, 2, 1))
        rows = cls(torch.arange(10), 10).view(10, 1) * 3 + cls(torch.arange(12), 12).view(1, 12)
        pre_b = pre + bias[:, rows].permute(0, 3, 1, 2)
        op([xb, hb], U, out2=RH, aux0=hb, img_bias=bias.contiguous(), img_bias_border=True)
        assert torch.allclose(U.to_nchw(), torch.sigmoid(pre_b[:, :ch]), **WTOL)
        assert torch.allclose(RH.to_nchw(), (1 - torch.sigmoid(pre_b[:, ch:])) * h, **WTOL)
    elif case == 'two sources -> GRU out':
        x, h = torch.randn(1, 32, 12, 15, generator=g), torch.randn(1, 64, 12, 15, generator=g)
        u = torch.rand(1, 64, 12, 15, generator=g)
        w = torch.randn(64, 96, 3, 3, generator=g) / 30
        sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
        op = ConvOp(sim, w, identity_chan_map(32) + identity_chan_map(64, offset=32), (4, 8), sc, sh, 'cpu', act=native.ACT_RELU,
                    epi=native.EPI_GRU_OUT, tune=True)
        xb, hb, ub = _to_buf(x), _to_buf(h), _to_buf(u)
        tilde = F.relu(F.conv2d(torch.cat([x, h], 1), w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        outs = check(op, lambda o, out: o([xb, hb], out, aux0=ub, aux1=hb), (1 - u) * h + u * tilde, (1, 12, 15, 64), 64)
        bias = torch.randn(1, 64, generator=g)                                # per-image bias rows (no border classes)
        tilde_b = F.relu((F.conv2d(torch.cat([x, h], 1), w, padding=1) + bias.view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        check(op, lambda o, out: o([xb, hb], out, aux0=ub, aux1=hb, img_bias=bias.contiguous()), (1 - u) * h + u * tilde_b, (1, 12, 15, 64), 64)
    elif case == '32->64 border-class bias':
        n, H, W = 2, 9, 8
        x = torch.randn(n, 32, H, W, generator=g)
        w = torch.randn(64, 32, 3, 3, generator=g) / 17
        bias = torch.randn(n, 9, 64, generator=g)
        op = ConvOp(sim, w, identity_chan_map(32), (4, 0), torch.ones(64), torch.zeros(64), 'cpu', act=native.ACT_RELU, tune=True)
        cls = lambda v, size: torch.where(v == 0, 0, torch.where(v == size - 1, 2, 1))
        rows = cls(torch.arange(H), H).view(H, 1) * 3 + cls(torch.arange(W), W).view(1, W)
        want = F.relu(F.conv2d(x, w, padding=1) + bias[:, rows].permute(0, 3, 1, 2))
        check(op, lambda o, out: o([_to_buf(x)], out, img_bias=bias.contiguous(), img_bias_border=True), want, (n, H, W, 64), 64)
    elif case == '16->256 sequence views':
        B, T, H, W = 2, 3, 6, 7
        seq = torch.randn(B * T, H, W, 16, generator=g)
        sbuf = Buf(seq, B * T, H, W, 16)
        w = torch.randn(256, 16, 3, 3, generator=g) * 0.1
        op = ConvOp(sim, w, identity_chan_map(16), (2, 0), torch.ones(256), torch.zeros(256), 'cpu', tune=True)
        dst = Buf.alloc(B * T, H, W, 256, 'cpu')
        op.force_form = wform
        op([sbuf.images(1, B, step=T)], dst.images(1, B, step=T))
        x_t = seq.view(B, T, H, W, 16)[:, 1].permute(0, 3, 1, 2)
        got = dst.nhwc().view(B, T, H, W, 256)[:, 1].permute(0, 3, 1, 2)
        assert torch.allclose(got, F.conv2d(x_t, w, padding=1), **WTOL)
        assert dst.nhwc().view(B, T, H, W, 256)[:, 0].abs().max() == 0
    else:
        x = torch.randn(2, 48, 7, 9, generator=g)
        w = torch.randn(40, 48, 3, 3, generator=g) / 20
        op = ConvOp(sim, w, identity_chan_map(48), (6, 0), torch.ones(40), torch.zeros(40), 'cpu', tune=True)
        out = Buf.alloc(2, 7, 9, 64, 'cpu')
        out.tensor.fill_(7.0)
        op.force_form = wform
        op([_to_buf(x)], out)
        assert torch.allclose(out.to_nchw()[:, :40], F.conv2d(x, w, padding=1), **WTOL)
        assert (out.to_nchw()[:, 40:] == 7.0).all(), 'padding couts are never stored'
