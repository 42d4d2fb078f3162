"""Warp and helper kernel *sources* on the CPU simulator versus the oracle / torch fp32 ops."""
import pytest
import torch
import torch.nn.functional as F

from fiery_amd import native
from oracle import bev_stack

TOL = dict(rtol=1e-5, atol=1e-5)


def test_warp_params_match_reference_pose_algebra(sim):
    g = torch.Generator().manual_seed(0)
    B, S = 3, 4
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 2.5 + torch.rand(B, S, generator=g)
    ego[..., 1] = 0.2 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.05 * torch.randn(B, S, generator=g)
    ego[..., 3:5] = 0.01 * torch.randn(B, S, 2, generator=g)      # small roll/pitch must be ignored consistently
    theta = sim.warp_params(ego, (50.0, 25.0))
    want = bev_stack.cumulative_warp_thetas(ego, (50.0, 25.0))
    for t in range(S - 1):
        assert torch.allclose(theta[:, t].view(B, 2, 3), want[t], **TOL)
    assert torch.equal(theta[:, S - 1], torch.tensor([1.0, 0, 0, 0, 1, 0]).expand(B, 6))


def test_warp_single_frame_is_identity(sim):
    theta = sim.warp_params(torch.randn(2, 1, 6), (50.0, 50.0))
    assert torch.equal(theta[:, 0], torch.tensor([1.0, 0, 0, 0, 1, 0]).expand(2, 6))


def test_bev_warp_matches_grid_sample_and_changes_layout(sim):
    g = torch.Generator().manual_seed(1)
    B, S, C, H, W = 2, 3, 5, 12, 70        # W > 64 exercises two x-tiles
    x = torch.randn(B, S, C, H, W, generator=g)
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 3.0 + torch.rand(B, S, generator=g)
    ego[..., 1] = 0.5 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.1 * torch.randn(B, S, generator=g)
    extent = (6.0, 35.0)
    want = bev_stack.cumulative_warp_features(x.clone(), ego, 'bilinear', extent)
    theta = sim.warp_params(ego, extent)
    out = torch.zeros(B * S, H, W, 8)
    identity = [(i % S) == S - 1 for i in range(B * S)]
    sim.bev_warp_nchw_to_nhwc(x.view(B * S, C, H, W), theta.view(B * S, 6), identity, out, 8, H * W * 8)
    got = out[..., :C].permute(0, 3, 1, 2).reshape(B, S, C, H, W)
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-5), (got - want).abs().max()
    assert torch.equal(got[:, -1], x[:, -1])          # the present frame is copied, not resampled
    assert out[..., C:].abs().max() == 0


@pytest.mark.parametrize('extent,hw', [((50.0, 50.0), (200, 200)), ((50.0, 25.0), (400, 200)), ((6.0, 35.0), (37, 53))])
def test_bev_warp_is_bitwise_aten_given_the_hosts_transforms(sim, extent, hw):
    """The resampling kernel rounds like ATen's CPU affine_grid + grid_sample (linspace halves, BLAS product with k
    ascending, fused un-normalisation, fused four-term sum: csrc/warp.hip, top): with the transforms of
    `host_warp_transforms` - the reference's own host operators - every element of every warped frame EQUALS the oracle's
    (= the reference's, tests/test_oracle_vs_reference.py), on white noise, at the real map sizes and at an odd one."""
    from fiery_amd.model import host_warp_transforms
    B, S, C = 2, 3, 3
    H, W = hw
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, S, C, H, W, generator=g)
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 2.5 + 0.5 * torch.rand(B, S, generator=g)
    ego[..., 1] = 0.1 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.02 * torch.randn(B, S, generator=g)
    ego[1, :, 2:5] = 0.01 * torch.randn(S, 3, generator=g)            # roll / pitch / tz must be ignored the same way
    want = bev_stack.cumulative_warp_features(x, ego, 'bilinear', extent)
    theta = host_warp_transforms(ego, extent)
    for t, wt in enumerate(bev_stack.cumulative_warp_thetas(ego, extent)):
        assert torch.equal(theta[:, t], wt.reshape(B, 6))
    out = torch.zeros(B * S, H, W, 8)
    identity = [(i % S) == S - 1 for i in range(B * S)]
    sim.bev_warp_nchw_to_nhwc(x.view(B * S, C, H, W), theta.view(B * S, 6).contiguous(), identity, out, 8, H * W * 8)
    got = out[..., :C].permute(0, 3, 1, 2).reshape(B, S, C, H, W)
    assert torch.equal(got, want), (got - want).abs().max()
    # the device's own pose algebra: equal for most transforms, one ulp off where MKL / SLEEF do not round correctly
    dev_theta = sim.warp_params(ego, extent)
    assert (dev_theta - theta).abs().max() <= 1e-8
    assert (dev_theta == theta).float().mean() > 0.7


def test_bev_warp_both_grid_product_forms(sim):
    """MKL multiplies affine_grid's base grid with theta^T fused on Intel hosts and with separate roundings on AMD hosts
    (measured on the MI355X box's EPYC, tools/probe/aten_warp_probe.py).  flags = 0 must equal grid_sample on a grid whose
    product was formed with separately rounded torch operations; the fused flag must equal the fma form; and
    `warp_flags_of_this_host` must pick the one this machine's affine_grid produces."""
    g = torch.Generator().manual_seed(8)
    n, C, H, W = 2, 2, 200, 200
    x = torch.randn(n, C, H, W, generator=g)
    theta = torch.tensor([[0.99981, -0.01937, 0.01234, 0.01937, 0.99981, -0.05678],
                          [0.9995, 0.0316, -0.0021, -0.0316, 0.9995, -0.0611]])
    xs = (torch.linspace(-1, 1, W) * (W - 1) / W).view(1, 1, W)
    ys = (torch.linspace(-1, 1, H) * (H - 1) / H).view(1, H, 1)
    t = theta.view(n, 6, 1, 1)
    first = [xs * t[:, 0], xs * t[:, 3]]
    plain = torch.stack([(first[0] + ys * t[:, 1]) + t[:, 2], (first[1] + ys * t[:, 4]) + t[:, 5]], dim=-1)
    fused = torch.stack([(ys.double() * t[:, 1].double() + first[0].double()).float() + t[:, 2],
                         (ys.double() * t[:, 4].double() + first[1].double()).float() + t[:, 5]], dim=-1)
    local = F.affine_grid(theta.view(n, 2, 3), (n, C, H, W), align_corners=False)
    flags = native.warp_flags_of_this_host(H, W)
    assert torch.equal(local, fused if flags else plain)
    for grid, fl in ((plain, 0), (fused, native.WARP_FUSED_GRID_PRODUCT)):
        want = F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
        out = torch.zeros(n, H, W, 8)
        sim.bev_warp_nchw_to_nhwc(x, theta, [False, False], out, 8, H * W * 8, flags=fl)
        assert torch.equal(out[..., :C].permute(0, 3, 1, 2), want)
        want = F.grid_sample(x, grid, mode='nearest', padding_mode='zeros', align_corners=False)
        assert torch.equal(sim.bev_warp_nearest(x, theta, flags=fl), want)


def test_host_warp_transforms_large_rotations_and_single_frame():
    from fiery_amd.model import host_warp_transforms
    g = torch.Generator().manual_seed(5)
    ego = torch.randn(4, 5, 6, generator=g)
    theta = host_warp_transforms(ego, (50.0, 25.0))
    for t, wt in enumerate(bev_stack.cumulative_warp_thetas(ego, (50.0, 25.0))):
        assert torch.equal(theta[:, t], wt.reshape(4, 6))
    assert torch.equal(theta[:, 4], torch.tensor([1.0, 0, 0, 0, 1, 0]).expand(4, 6))
    assert torch.equal(host_warp_transforms(ego[:, :1], (50.0, 50.0))[:, 0], torch.tensor([1.0, 0, 0, 0, 1, 0]).expand(4, 6))


@pytest.mark.parametrize('C', [70, 64])          # 70: scalar rows; 64 of 72: the 16-byte kernel
def test_spatial_mean(sim, C):
    g = torch.Generator().manual_seed(2)
    n, H, W = 3, 9, 11
    x = torch.randn(n, H, W, 72, generator=g)
    out = torch.empty(n, C)
    ws = torch.empty(n * C * 64)
    sim.spatial_mean(x, 72, H * W * 72, n, 0, 1, H * W, C, out, ws)
    assert torch.allclose(out, x[..., :C].mean(dim=(1, 2)), **TOL)
    # two-frame windows of a (batch=1, time=3) buffer: frames (0,1) and (1,2)
    out2 = torch.empty(2, C)
    sim.spatial_mean(x, 72, 0, 1, H * W * 72, 2, 2 * H * W, C, out2, ws)
    want = torch.stack([x[0:2, ..., :C].mean(dim=(0, 1, 2)), x[1:3, ..., :C].mean(dim=(0, 1, 2))])
    assert torch.allclose(out2, want, **TOL)


@pytest.mark.parametrize('act', [native.ACT_NONE, native.ACT_RELU])
def test_rowwise_dense(sim, act):
    g = torch.Generator().manual_seed(3)
    rows, n_in, n_out = 5, 6, 23
    v = torch.randn(rows, 8, generator=g)
    w = torch.randn(n_out, 70, generator=g)
    sc, sh = torch.rand(n_out, generator=g) + 0.5, torch.randn(n_out, generator=g)
    y = torch.zeros(rows, 24)
    sim.rowwise_dense(v, 8, rows, n_in, w, 70, 64, n_out, sc, sh, act, False, y, 24)
    want = (v[:, :n_in] @ w[:, 64:70].t()) * sc + sh
    if act == native.ACT_RELU:
        want = F.relu(want)
    assert torch.allclose(y[:, :n_out], want, **TOL)
    before = y.clone()
    sim.rowwise_dense(v, 8, rows, n_in, w, 70, 0, n_out, None, None, native.ACT_NONE, True, y, 24, w_mul=0.5)
    assert torch.allclose(y[:, :n_out], before[:, :n_out] + 0.5 * (v[:, :n_in] @ w[:, :n_in].t()), **TOL)
    sim.rowwise_dense(v, 8, rows, n_in, w, 70, 0, n_out, None, None, native.ACT_NONE, False, y, 24, lo=-0.25, hi=0.5)
    assert torch.allclose(y[:, :n_out], (v[:, :n_in] @ w[:, :n_in].t()).clamp(-0.25, 0.5), **TOL)


def test_latent_sample(sim):
    g = torch.Generator().manual_seed(9)
    mu, ls_, eps = torch.randn(3, 32, generator=g), torch.randn(3, 32, generator=g), torch.randn(3, 32, generator=g)
    out = torch.empty(3, 32)
    sim.latent_sample(mu, ls_, eps, 32, 3, 32, out, 32)
    assert torch.allclose(out, mu + torch.exp(ls_) * eps, **TOL)
    sim.latent_sample(mu, ls_, None, 32, 3, 32, out, 32)
    assert torch.equal(out, mu)


@pytest.mark.parametrize('hw', [(8, 10), (7, 13)])
def test_maxpool2x2_with_zero_padding_of_odd_sizes(sim, hw):
    g = torch.Generator().manual_seed(4)
    H, W = hw
    x = torch.randn(2, 16, H, W, generator=g) - 0.5
    xin = x.permute(0, 2, 3, 1).contiguous()
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    out = torch.empty(2, Ho, Wo, 16)
    sim.maxpool2x2(xin, 16, 2, H, W, 16, out, 16)
    want = F.max_pool2d(F.pad(x, (0, W % 2, 0, H % 2), value=0), 2, 2)
    assert torch.equal(out.permute(0, 3, 1, 2), want)


@pytest.mark.parametrize('C', [16, 10])          # 16: the 16-byte kernel; 10: scalar
def test_upsample2x_add(sim, C):
    g = torch.Generator().manual_seed(5)
    n, H, W = 2, 5, 7
    x = torch.randn(n, C, H, W, generator=g)
    skip = torch.randn(n, C, 2 * H, 2 * W, generator=g)
    shift = torch.randn(C, generator=g)
    out = torch.empty(n, 2 * H, 2 * W, C)
    sim.upsample2x_add(x.permute(0, 2, 3, 1).contiguous(), C, n, H, W, C, shift,
                       skip.permute(0, 2, 3, 1).contiguous(), C, out, C)
    want = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) + shift.view(1, -1, 1, 1) + skip
    assert torch.allclose(out.permute(0, 3, 1, 2), want, **TOL)


def test_broadcast_and_layout_changes(sim):
    g = torch.Generator().manual_seed(6)
    n, C, HW = 2, 32, 75
    v = torch.randn(n, C, generator=g)
    out = torch.zeros(n, HW, 40)
    sim.broadcast(v, C, n, HW, C, out, 40, HW * 40)
    assert torch.equal(out[..., :C], v.view(n, 1, C).expand(n, HW, C))
    assert out[..., C:].abs().max() == 0
    x = torch.randn(n, 7, HW, generator=g)
    nhwc = torch.zeros(n, HW, 8)
    sim.nchw_to_nhwc(x, n, 7, HW, nhwc, 8, HW * 8)
    assert torch.equal(nhwc[..., :7], x.permute(0, 2, 1))
    back = torch.empty(n, 7, HW)
    sim.nhwc_to_nchw(nhwc, 8, HW * 8, n, 7, HW, back)
    assert torch.equal(back, x)


def test_sequential_window_mean_reproduces_aten_avg_pool3d_on_constant_channels(sim):
    """ATen's avg_pool3d adds the window's elements one by one in fp32; for the spatially constant ego-pose
    channels that is a systematic rounding drift the reference's numbers contain.  Must match bit for bit."""
    H, W = 200, 200                                                 # the real baseline.yml window: 2 x 40,000 additions
    g = torch.Generator().manual_seed(4)
    prev = torch.randn(8, 6, generator=g) * 2.5
    cur = torch.randn(8, 6, generator=g) * 2.5
    prev[0] = 0.0                                                   # fiery.py:152-154: zeros at t = 0
    cur[1] = torch.tensor([0.5, -0.25, 3.0, 1e-9, 0.0, 2.5])        # exact ties, stagnation, zero
    prev[2], cur[2] = torch.full((6,), 0.02), torch.full((6,), -0.021)    # sign change half way
    out = torch.empty(8, 6)
    sim.sequential_window_mean(prev, cur, 8, 6, H * W, out, 6)
    x = torch.stack([prev, cur], dim=2).view(8, 6, 2, 1, 1).expand(8, 6, 2, H, W).contiguous()
    want = F.avg_pool3d(x, kernel_size=(2, H, W), stride=(1, H, W), padding=(1, 0, 0), count_include_pad=False)[:, :, 1, 0, 0]
    assert torch.equal(out, want)
    sim.sequential_window_mean(None, prev, 8, 6, H * W, out, 6)     # window clipped at t = 0: first frame only
    want0 = F.avg_pool3d(x, kernel_size=(2, H, W), stride=(1, H, W), padding=(1, 0, 0), count_include_pad=False)[:, :, 0, 0, 0]
    assert torch.equal(out, want0)


# ------------------------------------------------------------------------------------------------------
# image-trunk helpers: depthwise convolution, squeeze-and-excite scaling, swish
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('k,stride,pads', [(3, 1, (1, 1, 1, 1)), (5, 1, (2, 2, 2, 2)), (3, 2, (0, 1, 0, 1)), (5, 2, (1, 2, 1, 2)),
                                           (3, 2, (1, 1, 1, 1))])
@pytest.mark.parametrize('tiled', ['1', '0'])
def test_depthwise_conv_bn_swish(sim, monkeypatch, k, stride, pads, tiled):
    """efficientnet-pytorch's `_depthwise_conv` (static 'same' zero padding, asymmetric when the stride is 2) + `_bn1` +
    swish, against torch."""
    from fiery_amd import native
    monkeypatch.setenv('FIERY_DEPTHWISE_TILED', tiled)          # the register-tiled kernels and the general one
    g = torch.Generator().manual_seed(k * 10 + stride)
    n, C, H, W = 2, 12, 9, 11
    x = torch.randn(n, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    left, right, top, bottom = pads
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, pads), w, stride=stride, groups=C)
    ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = ref * torch.sigmoid(ref)
    Ho, Wo = ref.shape[-2:]
    xin = torch.zeros(n, H, W, 16)
    xin[..., :C] = x.permute(0, 2, 3, 1)
    out = torch.full((n, Ho, Wo, 16), float('nan'))
    wt = torch.zeros(k * k, 12)
    wt[:, :C] = w.view(C, k * k).t()
    sim.depthwise_conv(xin, 16, n, H, W, C, wt.contiguous(), 12, k, stride, top, left, Ho, Wo, scale, shift, native.ACT_SWISH, out, 16)
    got = out[..., :C].permute(0, 3, 1, 2)
    assert (got - ref).abs().max() < 1e-5
    assert torch.isnan(out[..., C:]).all()                       # only the C channels asked for are written


def test_scale_channels_and_swish_dense(sim):
    from fiery_amd import native
    g = torch.Generator().manual_seed(3)
    n, HW, C = 3, 35, 8
    x = torch.randn(n, HW, 12, generator=g)
    gate = torch.rand(n, 8, generator=g)
    want = x.clone()
    want[..., :C] *= gate.view(n, 1, C)
    sim.scale_channels(x, 12, n, HW, C, gate, 8)
    assert torch.equal(x, want)
    # the squeeze-and-excite MLP: swish(W1 v + b1), sigmoid(W2 . + b2)
    v = torch.randn(n, C, generator=g)
    w1, b1 = torch.randn(3, C, generator=g), torch.randn(3, generator=g)
    h = torch.empty(n, 3)
    sim.rowwise_dense(v, C, n, C, w1, C, 0, 3, None, b1, native.ACT_SWISH, False, h, 3)
    ref = v @ w1.t() + b1
    assert torch.allclose(h, ref * torch.sigmoid(ref), atol=1e-6)


def test_conv_swish_epilogue(sim):
    """1x1 expansion conv + folded BN + swish (MBConv `_expand_conv` + `_bn0` + swish)."""
    from fiery_amd import native
    from fiery_amd.ops import Buf, ConvOp, identity_chan_map
    g = torch.Generator().manual_seed(4)
    n, H, W, cin, cout = 2, 5, 7, 8, 40
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * 0.3
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    op = ConvOp(sim, w, identity_chan_map(cin), (1, 0), scale, shift, 'cpu', act=native.ACT_SWISH)
    src = Buf.alloc(n, H, W, cin, 'cpu')
    src.nhwc()[..., :cin] = x.permute(0, 2, 3, 1)
    dst = Buf.alloc(n, H, W, cout, 'cpu')
    op([src], dst)
    ref = torch.nn.functional.conv2d(x, w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    ref = ref * torch.sigmoid(ref)
    assert (dst.to_nchw()[:, :cout] - ref).abs().max() < 1e-5


@pytest.mark.parametrize('C,sq', [(48, 12), (960, 40), (20, 1)])
def test_se_gate(sim, C, sq):
    g = torch.Generator().manual_seed(C + sq)
    n = 3
    mean = torch.randn(n, C + 4, generator=g)
    w1, b1 = torch.randn(sq, C, generator=g) * 0.2, torch.randn(sq, generator=g)
    w2, b2 = torch.randn(C, sq, generator=g) * 0.2, torch.randn(C, generator=g)
    gate = torch.full((n, C + 4), float('nan'))
    sim.se_gate(mean, C + 4, n, C, w1, b1, sq, w2, b2, gate, C + 4)
    h = mean[:, :C] @ w1.t() + b1
    h = h * torch.sigmoid(h)
    want = torch.sigmoid(h @ w2.t() + b2)
    assert torch.allclose(gate[:, :C], want, atol=2e-6)
    assert torch.isnan(gate[:, C:]).all()


def test_se_gate_from_the_feature_map(sim):
    g = torch.Generator().manual_seed(9)
    n, H, W, C, sq = 2, 7, 9, 24, 6
    x = torch.randn(n, H, W, 32, generator=g)
    w1, b1 = torch.randn(sq, C, generator=g) * 0.3, torch.randn(sq, generator=g)
    w2, b2 = torch.randn(C, sq, generator=g) * 0.3, torch.randn(C, generator=g)
    gate = torch.zeros(n, 24)
    ws = torch.zeros(n * C * 64)
    sim.se_gate_nhwc(x, 32, H * W * 32, n, H * W, C, w1, b1, sq, w2, b2, gate, 24, ws)
    mean = x[..., :C].mean(dim=(1, 2))
    h = mean @ w1.t() + b1
    h = h * torch.sigmoid(h)
    assert torch.allclose(gate, torch.sigmoid(h @ w2.t() + b2), atol=2e-6)


# ------------------------------------------------------------------------------------------------------
# the step after the path: per-frame instance segmentation (reference: fiery/utils/instance.py:80-144)
# ------------------------------------------------------------------------------------------------------
def _instance_case(seed, H=40, W=56, n_blobs=7, all_foreground=False):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float), torch.arange(W, dtype=torch.float), indexing='ij')
    center = torch.zeros(H, W)
    cy, cx = torch.rand(n_blobs, generator=g) * H, torch.rand(n_blobs, generator=g) * W
    for k in range(n_blobs):
        center = torch.maximum(center, torch.exp(-((yy - cy[k]) ** 2 + (xx - cx[k]) ** 2) / 9.0))
    center = center + 0.01 * torch.rand(H, W, generator=g)
    nearest = torch.stack([(yy - cy[k]) ** 2 + (xx - cx[k]) ** 2 for k in range(n_blobs)]).argmin(0)
    offset = torch.stack([cy[nearest] - yy, cx[nearest] - xx]) + 0.3 * torch.randn(2, H, W, generator=g)
    fg = torch.ones(H, W, dtype=torch.bool) if all_foreground else center > 0.05
    return center, offset, fg


@pytest.mark.parametrize('seed,kw', [(0, {}), (1, dict(n_blobs=1)), (2, dict(all_foreground=True)),
                                     (3, dict(n_blobs=140, H=64, W=64)), (4, dict(H=33, W=70, n_blobs=12))])
def test_instance_segmentation_matches_the_oracle(sim, seed, kw):
    """Centres bit for bit (index work); ids bit for bit too on these seeds (a pixel could only differ if its two
    nearest centres were equidistant to within an ulp); more than 100 centres, a frame without background, one blob."""
    from fiery_amd import instance as hip_instance
    from oracle import instance as oi
    center, offset, fg = _instance_case(seed, **kw)
    want_seg, want_centers = oi.instance_segmentation_and_centers(center, offset, fg)
    got_seg, got_centers = hip_instance.get_instance_segmentation_and_centers(center, offset, fg, lib=sim)
    assert torch.equal(got_centers, want_centers)
    assert got_seg.shape == want_seg.shape and got_seg.dtype == want_seg.dtype
    assert torch.equal(got_seg, want_seg)


def test_instance_segmentation_batched_frames_and_empty_frame(sim):
    from fiery_amd import instance as hip_instance
    from oracle import instance as oi
    cases = [_instance_case(s, H=24, W=40, n_blobs=4) for s in (5, 6, 7)]
    center = torch.stack([c for c, _, _ in cases])
    center[1] = 0.0                                                  # nothing above the threshold in frame 1
    offset = torch.stack([o for _, o, _ in cases])
    fg = torch.stack([m for _, _, m in cases])
    seg, centers, count = hip_instance.instance_segmentation_frames(center, offset, fg, lib=sim)
    for f in range(3):
        want_seg, want_centers = oi.instance_segmentation_and_centers(center[f], offset[f], fg[f])
        assert torch.equal(seg[f:f + 1], want_seg)
        assert int(count[f]) == len(want_centers)
        assert torch.equal(centers[f, :len(want_centers)], want_centers.long()) and (centers[f, len(want_centers):] == -1).all()
    assert int(count[1]) == 0 and seg[1].abs().sum() == 0
    with pytest.raises(ValueError):
        hip_instance.instance_segmentation_frames(center, offset, fg, nms_kernel_size=5, lib=sim)


def test_reverse_warp_params_and_nearest_warp_match_the_oracle(sim):
    """`fiery_warp_params_reverse` + `fiery_bev_warp_nearest_nchw` (label warping, trainer.py:133-191) against the oracle's
    `cumulative_warp_features_reverse` (bitwise equal to the reference's, tests/test_oracle_vs_reference.py)."""
    g = torch.Generator().manual_seed(8)
    B, S, C, H, W = 2, 5, 3, 20, 28
    x = torch.randn(B, S, C, H, W, generator=g)
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 0.8 + 0.4 * torch.rand(B, S, generator=g)
    ego[..., 1] = 0.2 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.05 * torch.randn(B, S, generator=g)
    ego[..., 3:5] = 0.01 * torch.randn(B, S, 2, generator=g)
    extent = (7.0, 10.0)
    theta = sim.warp_params_reverse(ego, extent)
    want_theta = bev_stack.cumulative_warp_reverse_thetas(ego, extent)
    assert torch.equal(theta[:, 0], torch.tensor([1.0, 0, 0, 0, 1, 0]).expand(B, 6))
    for i in range(1, S):
        assert torch.allclose(theta[:, i].view(B, 2, 3), want_theta[i - 1], **TOL)
    got = sim.bev_warp_nearest(x.view(B * S, C, H, W).contiguous(), theta.view(B * S, 6)).view(B, S, C, H, W)
    want = bev_stack.cumulative_warp_features_reverse(x, ego, 'nearest', extent)
    assert torch.equal(got[:, 0], x[:, 0])
    mismatch = (got != want).any(dim=2).float().mean().item()          # pixels whose nearest source differs (boundary ties)
    assert mismatch < 5e-3, mismatch
    assert (want[:, 1:] == 0).float().mean() > 0.01                    # some of the map leaves the grid: zeros padding


@pytest.mark.parametrize('shape', [(2, 8, 5, 7), (1, 4, 1, 1), (1, 12, 1, 6), (2, 4, 3, 1)])
def test_upsample2x_backward_is_the_transpose_of_the_interpolation(sim, shape):
    """`fiery_upsample2x_bwd_nhwc` against autograd through F.interpolate (borders included: one-pixel rows/columns)."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * 10 + w)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    gy = torch.randn(up.shape, generator=g)
    (want,) = torch.autograd.grad(up, x, gy)
    got = sim.upsample2x_bwd(gy.permute(0, 2, 3, 1).contiguous(), n, h, w, c).permute(0, 3, 1, 2)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)


def _label_blobs(seed, T, H, W, n_obj):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(T, H, W, dtype=torch.int64)
    centres = torch.rand(n_obj, 2, generator=g) * torch.tensor([H - 10.0, W - 10.0]) + 5.0
    speed = torch.randn(n_obj, 2, generator=g) * 1.5
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    for t in range(T):
        for k in range(n_obj):
            if (k + t + seed) % 5 == 0:
                continue
            c = centres[k] + speed[k] * t
            ids[t][((yy - c[0]).abs() < 2.5 + k % 3) & ((xx - c[1]).abs() < 3.5)] = k + 1
    ego = torch.zeros(T, 6)
    ego[:, 0] = 1.0 + torch.rand(T, generator=g)
    ego[:, 1] = 0.3 * torch.randn(T, generator=g)
    ego[:, 5] = 0.05 * torch.randn(T, generator=g)
    return ids, ego


@pytest.mark.parametrize('seed,T,hw,n_obj', [(0, 4, (32, 48), 6), (3, 1, (20, 20), 3), (5, 6, (50, 50), 40)])
def test_instance_labels_against_the_oracle(sim, seed, T, hw, n_obj):
    """`fiery_instance_labels` (+ the nearest warp of the id maps) against oracle/labels.py - itself pinned to the reference's
    `convert_instance_mask_to_center_and_offset_label` -: integer-valued outputs exactly, the heat map to an ulp of exp."""
    from fiery_amd.labels import convert_instance_mask_to_center_and_offset_label
    from oracle.labels import instance_labels
    ids, ego = _label_blobs(seed, T, *hw, n_obj)
    extent = (hw[0] / 2.0, hw[1] / 2.0)
    want = instance_labels(ids, ego, n_obj, 255, 3, extent)
    got = convert_instance_mask_to_center_and_offset_label(ids, ego, n_obj, ignore_index=255, spatial_extent=extent, lib=sim, device='cpu')
    assert torch.allclose(got[0], want[0], rtol=0, atol=2e-7)
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])


@pytest.mark.parametrize('in_hw,scale,final,top_crop', [((90, 160), 0.3, (20, 44), 5), ((45, 80), 0.7, (28, 60), 2), ((64, 64), 1.0, (64, 64), 0),
                                                        ((30, 50), 1.6, (40, 70), 4), ((40, 72), 0.5, (24, 40), 3)])
def test_image_resize_crop_normalise_is_pillow_byte_for_byte(sim, in_hw, scale, final, top_crop):
    """`fiery_image_resize_crop_normalise` + the host coefficient tables against the real Pillow resize / crop and the
    ToTensor + Normalize restatement: down-scaling (antialiased), up-scaling, identity, and a crop window that reaches past the
    resized image (Pillow pads with black).  The normalised floats must be equal bit for bit."""
    from fiery_amd.images import resize_crop_normalise
    from oracle.images import prepare
    g = torch.Generator().manual_seed(in_hw[0] * 7 + in_hw[1])
    images = torch.randint(0, 256, (3, in_hw[0], in_hw[1], 3), generator=g, dtype=torch.uint8)
    images[0, ::2] = 255                                       # hard edges: every tap's rounding matters
    resize_dims = (int(in_hw[1] * scale), int(in_hw[0] * scale))
    left = int(max(0, (resize_dims[0] - final[1]) / 2))
    crop = (left, top_crop, left + final[1], top_crop + final[0])
    want = prepare(images.numpy(), resize_dims, crop)
    got = resize_crop_normalise(images, resize_dims, crop, lib=sim, device='cpu')
    assert got.shape == want.shape
    assert torch.equal(got, want), (got - want).abs().max()


@pytest.mark.parametrize('preset', ['baseline.yml', 'lyft/baseline.yml'])
def test_image_preparation_with_the_presets_own_sizes(sim, preset):
    """One frame at the dataset's original size through the preset's own resize / crop parameters (nuScenes 900 x 1600 x 0.3,
    Lyft 1080 x 1920 x 0.25 - whatever the YAML says) on the simulated kernels: Pillow's bytes, and the intrinsics update as the
    reference computes it."""
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.images import get_resizing_and_cropping_parameters, resize_crop_normalise, update_intrinsics
    from oracle.images import prepare
    cfg = get_preset_cfg(preset)
    aug = get_resizing_and_cropping_parameters(cfg)
    h, w = cfg.IMAGE.ORIGINAL_HEIGHT, cfg.IMAGE.ORIGINAL_WIDTH
    g = torch.Generator().manual_seed(h)
    low = torch.randint(0, 256, (1, h // 8 + 1, w // 8 + 1, 3), generator=g, dtype=torch.uint8)
    image = low.repeat_interleave(8, dim=1).repeat_interleave(8, dim=2)[:, :h, :w].contiguous()
    image[:, ::5, ::3] = torch.randint(0, 256, image[:, ::5, ::3].shape, generator=g, dtype=torch.uint8)
    want = prepare(image.numpy(), aug['resize_dims'], aug['crop'])
    got = resize_crop_normalise(image, aug['resize_dims'], aug['crop'], lib=sim, device='cpu')
    assert tuple(got.shape[-2:]) == tuple(cfg.IMAGE.FINAL_DIM)
    assert torch.equal(got, want)
    K = torch.tensor([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    new = update_intrinsics(K, aug['crop'][1], aug['crop'][0], scale_width=aug['scale_width'], scale_height=aug['scale_height'])
    s = cfg.IMAGE.RESIZE_SCALE
    assert torch.allclose(new, torch.tensor([[1266.4 * s, 0.0, 816.3 * s - aug['crop'][0]], [0.0, 1266.4 * s, 491.5 * s - aug['crop'][1]],
                                             [0.0, 0.0, 1.0]]))
