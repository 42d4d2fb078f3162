"""CPU restatement of the BEV stack downstream of voxel pooling.  TEST INFRASTRUCTURE.

Functional PyTorch-CPU fp32 (this is floating-point convolution work, so the checker is a plain
torch fp32 restatement), driven by a flat `state_dict` with the reference's key names.  Each function
cites the reference lines it follows.  Inference semantics only (`model.eval()`): BatchNorm uses
running statistics, Dropout2d(p=0) is the identity, the latent sample is
`present_mu + exp(present_log_sigma) * noise`.

Pinned against the reference modules executed in the build container
(tests/test_oracle_vs_reference.py) and against tests/golden/*.npz.
"""
import math

import torch
import torch.nn.functional as F

from . import lift_splat

BN_EPS = 1e-5


class Weights:
    """Prefix-scoped view of a flat state_dict."""

    def __init__(self, state_dict, prefix=''):
        self.sd, self.prefix = state_dict, prefix

    def sub(self, name):
        return Weights(self.sd, f'{self.prefix}{name}.')

    def has(self, name):
        return f'{self.prefix}{name}' in self.sd

    def __getitem__(self, name):
        return self.sd[f'{self.prefix}{name}']

    def get(self, name):
        return self.sd.get(f'{self.prefix}{name}')


def _bn(x, w):
    return F.batch_norm(x, w['running_mean'], w['running_var'], w['weight'], w['bias'], False, 0.0, BN_EPS)


# ---------------------------------------------------------------------------------------------
# ego-motion warping.  reference: fiery/utils/geometry.py:82-157, 181-253
# ---------------------------------------------------------------------------------------------
def euler_to_matrix(angle):
    """fiery/utils/geometry.py:109-140: R = Rx . Ry . Rz."""
    shape = angle.shape
    a = angle.reshape(-1, 3)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    zeros, ones = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zeros, sz, cz, zeros, zeros, zeros, ones], dim=1).view(-1, 3, 3)
    ymat = torch.stack([cy, zeros, sy, zeros, ones, zeros, -sy, zeros, cy], dim=1).view(-1, 3, 3)
    xmat = torch.stack([ones, zeros, zeros, zeros, cx, -sx, zeros, sx, cx], dim=1).view(-1, 3, 3)
    return xmat.bmm(ymat).bmm(zmat).view(*shape[:-1], 3, 3)


def pose_to_matrix(vec):
    """fiery/utils/geometry.py:143-157."""
    rot = euler_to_matrix(vec[..., 3:].contiguous())
    mat = torch.cat([rot, vec[..., :3].unsqueeze(-1)], dim=-1)
    mat = F.pad(mat, [0, 0, 0, 1], value=0)
    mat[..., 3, 3] = 1.0
    return mat


def matrix_to_pose(matrix):
    """fiery/utils/geometry.py:82-106."""
    rotx = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    cosy = torch.sqrt(matrix[..., 1, 2] ** 2 + matrix[..., 2, 2] ** 2)
    roty = torch.atan2(matrix[..., 0, 2], cosy)
    rotz = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rotx, roty, rotz), dim=-1)), dim=-1)


def warp_affine_params(flow, spatial_extent):
    """2x3 sampling transform of `warp_features` (fiery/utils/geometry.py:192-215) from a 6-DoF vector."""
    angle = flow[:, 5]
    tx = -(flow[:, 0] / spatial_extent[0])
    ty = flow[:, 1] / spatial_extent[1]
    c, s = torch.cos(angle), torch.sin(angle)
    return torch.stack([c, -s, ty, s, c, tx], dim=-1).view(-1, 2, 3)


def warp_features(x, flow, mode, spatial_extent):
    """fiery/utils/geometry.py:181-222 (the reference samples with `grid.float()`; `.to(x.dtype)` is the same
    thing for its fp32 tensors and lets the float64 evaluation below go through)."""
    theta = warp_affine_params(flow, spatial_extent)
    grid = F.affine_grid(theta, size=x.shape, align_corners=False)
    return F.grid_sample(x, grid.to(x.dtype), mode=mode, padding_mode='zeros', align_corners=False)


def cumulative_warp_thetas(flow, spatial_extent):
    """Sampling transforms for frames 0..S-2 (the last frame is never warped).
    fiery/utils/geometry.py:225-253: frame t uses flow[t] @ ... @ flow[S-2]."""
    s = flow.shape[1]
    mats = pose_to_matrix(flow)
    thetas = [None] * (s - 1)
    cum = mats[:, -2]
    for t in reversed(range(s - 1)):
        thetas[t] = warp_affine_params(matrix_to_pose(cum), spatial_extent)
        cum = mats[:, t - 1] @ cum
    return thetas


def cumulative_warp_features(x, flow, mode, spatial_extent):
    """fiery/utils/geometry.py:225-253."""
    s = x.shape[1]
    if s == 1:
        return x
    mats = pose_to_matrix(flow)
    out = [x[:, -1]]
    cum = mats[:, -2]
    for t in reversed(range(s - 1)):
        out.append(warp_features(x[:, t], matrix_to_pose(cum), mode, spatial_extent))
        cum = mats[:, t - 1] @ cum
    return torch.stack(out[::-1], 1)


def invert_pose_matrix(x):
    """[B, 4, 4] pose matrices -> their inverses [R^T | -R^T t] (fiery/utils/geometry.py:160-178)."""
    rt = x[:, :3, :3].transpose(1, 2)
    inv = torch.cat([rt, -torch.bmm(rt, x[:, :3, 3:])], dim=-1)
    inv = F.pad(inv, [0, 0, 0, 1], value=0)
    inv[..., 3, 3] = 1.0
    return inv


def cumulative_warp_features_reverse(x, flow, mode, spatial_extent):
    """fiery/utils/geometry.py:256-280: frame 0 unchanged, frame i warped by inverse(flow[0]) @ ... @ inverse(flow[i-1])."""
    mats = pose_to_matrix(flow)
    out = [x[:, 0]]
    cum = None
    for i in range(1, x.shape[1]):
        cum = invert_pose_matrix(mats[:, 0]) if i == 1 else cum @ invert_pose_matrix(mats[:, i - 1])
        out.append(warp_features(x[:, i], matrix_to_pose(cum), mode, spatial_extent))
    return torch.stack(out, 1)


def cumulative_warp_reverse_thetas(flow, spatial_extent):
    """The sampling transforms of frames 1 .. S-1 of `cumulative_warp_features_reverse`."""
    mats = pose_to_matrix(flow)
    thetas, cum = [], None
    for i in range(1, flow.shape[1]):
        cum = invert_pose_matrix(mats[:, 0]) if i == 1 else cum @ invert_pose_matrix(mats[:, i - 1])
        thetas.append(warp_affine_params(matrix_to_pose(cum), spatial_extent))
    return thetas


# ---------------------------------------------------------------------------------------------
# building blocks.  reference: fiery/layers/convolutions.py, fiery/layers/temporal.py
# ---------------------------------------------------------------------------------------------
def bottleneck(x, w, downsample=False):
    """fiery/layers/convolutions.py:64-168 (Dropout2d(p=0) omitted: identity)."""
    lw = w.sub('layers')
    r = F.relu(_bn(F.conv2d(x, lw['conv_down_project.weight']), lw.sub('abn_down_project.0')))
    r = F.relu(_bn(F.conv2d(r, lw['conv.weight'], stride=2 if downsample else 1, padding=1), lw.sub('abn.0')))
    r = F.relu(_bn(F.conv2d(r, lw['conv_up_project.weight']), lw.sub('abn_up_project.0')))
    if w.has('projection.conv_skip_proj.weight'):
        pw = w.sub('projection')
        skip = x
        if downsample:
            skip = F.pad(skip, (0, skip.shape[-1] % 2, 0, skip.shape[-2] % 2), value=0)
            skip = F.max_pool2d(skip, 2, 2)
        skip = _bn(F.conv2d(skip, pw['conv_skip_proj.weight']), pw.sub('bn_skip_proj'))
        return r + skip
    return r + x


def _pointwise3d(x, w):
    """conv 1x1x1 + BN + ReLU.  fiery/layers/temporal.py:107-117."""
    return F.relu(_bn(F.conv3d(x, w['conv.weight']), w.sub('norm')))


def _causal3d(x, w):
    """fiery/layers/temporal.py:65-85: left time padding, 'same' spatial padding."""
    kt, kh, kw = w['conv.weight'].shape[2:]
    x = F.pad(x, ((kw - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kh - 1) // 2, kt - 1, 0))
    return F.relu(_bn(F.conv3d(x, w['conv.weight']), w.sub('norm')))


def temporal_block(x, w):
    """fiery/layers/temporal.py:218-281, x is (B, C, T, H, W)."""
    b, _, t, h, wd = x.shape
    paths = []
    for i in range(2):
        pw = w.sub(f'convolution_paths.{i}')
        paths.append(_causal3d(_pointwise3d(x, pw.sub('0')), pw.sub('1')))
    paths.append(_pointwise3d(x, w.sub('convolution_paths.2')))
    res = torch.cat(paths, dim=1)
    if w.has('pyramid_pooling.features.0.conv_bn_relu.conv.weight'):
        # fiery/layers/temporal.py:167-215 with pool_sizes=[(2,h,w)] (fiery/models/temporal_model.py:23)
        fw = w.sub('pyramid_pooling.features.0.conv_bn_relu')
        pooled = F.avg_pool3d(x, kernel_size=(2, h, wd), stride=(1, h, wd), padding=(1, 0, 0), count_include_pad=False)
        pooled = _pointwise3d(pooled, fw)[:, :, :-1].contiguous()
        c = pooled.shape[1]
        pooled = F.interpolate(pooled.view(b * t, c, *pooled.shape[-2:]), (h, wd), mode='bilinear', align_corners=False)
        res = torch.cat([res, pooled.view(b, c, t, h, wd)], dim=1)
    res = _pointwise3d(res, w.sub('aggregation.0'))
    if w.has('projection.0.weight'):
        x = _bn(F.conv3d(x, w['projection.0.weight']), w.sub('projection.1'))
    return x + res


def bottleneck3d(x, w):
    """fiery/layers/temporal.py:120-164."""
    lw = w.sub('layers')
    r = _pointwise3d(x, lw.sub('conv_down_project'))
    r = _causal3d(r, lw.sub('conv'))
    r = _pointwise3d(r, lw.sub('conv_up_project'))
    if w.has('projection.0.weight'):
        x = _bn(F.conv3d(x, w['projection.0.weight']), w.sub('projection.1'))
    return r + x


def temporal_model(x, w, receptive_field, n_inbetween=0):
    """fiery/models/temporal_model.py:47-52, x is (B, T, C, H, W)."""
    x = x.permute(0, 2, 1, 3, 4)
    mw = w.sub('model')
    idx = 0
    for _ in range(receptive_field - 1):
        x = temporal_block(x, mw.sub(str(idx)))
        idx += 1
        for _ in range(n_inbetween):
            x = bottleneck3d(x, mw.sub(str(idx)))
            idx += 1
    x = x.permute(0, 2, 1, 3, 4).contiguous()
    return x[:, (receptive_field - 1):]


def gru_cell(x, state, w, gru_bias_init=0.0):
    """fiery/layers/temporal.py:49-62 (note (1 - reset) * state)."""
    xs = torch.cat([x, state], dim=1)
    update = torch.sigmoid(F.conv2d(xs, w['conv_update.weight'], w['conv_update.bias'], padding=1) + gru_bias_init)
    reset = torch.sigmoid(F.conv2d(xs, w['conv_reset.weight'], w['conv_reset.bias'], padding=1) + gru_bias_init)
    tw = w.sub('conv_state_tilde')
    tilde = F.relu(_bn(F.conv2d(torch.cat([x, (1.0 - reset) * state], dim=1), tw['conv.weight'], padding=1),
                       tw.sub('norm')))
    return (1.0 - update) * state + update * tilde


def spatial_gru(x, state, w):
    """fiery/layers/temporal.py:27-47, x is (B, T, C, H, W)."""
    outs = []
    for t in range(x.shape[1]):
        state = gru_cell(x[:, t], state, w)
        outs.append(state)
    return torch.stack(outs, dim=1)


def future_prediction(x, hidden, w, n_gru_blocks, n_res_layers):
    """fiery/models/future_prediction.py:27-36."""
    for i in range(n_gru_blocks):
        x = spatial_gru(x, hidden, w.sub(f'spatial_grus.{i}'))
        b, n, c, h, wd = x.shape
        y = x.reshape(b * n, c, h, wd)
        for j in range(n_res_layers):
            y = bottleneck(y, w.sub(f'res_blocks.{i}.{j}'))
        x = y.view(b, n, c, h, wd)
    return x


def distribution(s_t, w, latent_dim, min_log_sigma, max_log_sigma):
    """fiery/models/distributions.py:28-39 and :52-56."""
    b = s_t.shape[0]
    y = s_t[:, 0]
    for i in range(4):
        y = bottleneck(y, w.sub(f'encoder.model.{i}'), downsample=True)
    y = F.adaptive_avg_pool2d(y, 1)
    y = F.conv2d(y, w['last_conv.1.weight'], w['last_conv.1.bias']).view(b, 1, 2 * latent_dim)
    mu, log_sigma = y[:, :, :latent_dim], y[:, :, latent_dim:]
    return mu, torch.clamp(log_sigma, min_log_sigma, max_log_sigma)


def _basic_block(x, w):
    """torchvision 0.8.1 resnet BasicBlock (third-party; used by fiery/models/decoder.py:10-17)."""
    stride = 2 if w.has('downsample.0.weight') else 1
    out = F.relu(_bn(F.conv2d(x, w['conv1.weight'], stride=stride, padding=1), w.sub('bn1')))
    out = _bn(F.conv2d(out, w['conv2.weight'], padding=1), w.sub('bn2'))
    if w.has('downsample.0.weight'):
        x = _bn(F.conv2d(x, w['downsample.0.weight'], stride=stride), w.sub('downsample.1'))
    return F.relu(out + x)


def _upsample_add(x, skip, w):
    """fiery/layers/convolutions.py:203-214."""
    x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    x = _bn(F.conv2d(x, w['upsample_layer.1.weight']), w.sub('upsample_layer.2'))
    return x + skip


def _head(x, w, sigmoid=False):
    y = F.relu(_bn(F.conv2d(x, w['0.weight'], padding=1), w.sub('1')))
    y = F.conv2d(y, w['3.weight'], w['3.bias'])
    return torch.sigmoid(y) if sigmoid else y


def decoder(x, w):
    """fiery/models/decoder.py:53-91, x is (B, S, C, H, W)."""
    b, s, c, h, wd = x.shape
    x = x.reshape(b * s, c, h, wd)
    skip1 = x
    x = F.relu(_bn(F.conv2d(x, w['first_conv.weight'], stride=2, padding=3), w.sub('bn1')))
    for i in range(2):
        x = _basic_block(x, w.sub(f'layer1.{i}'))
    skip2 = x
    for i in range(2):
        x = _basic_block(x, w.sub(f'layer2.{i}'))
    skip3 = x
    for i in range(2):
        x = _basic_block(x, w.sub(f'layer3.{i}'))
    x = _upsample_add(x, skip3, w.sub('up3_skip'))
    x = _upsample_add(x, skip2, w.sub('up2_skip'))
    x = _upsample_add(x, skip1, w.sub('up1_skip'))
    out = {
        'segmentation': _head(x, w.sub('segmentation_head')),
        'instance_center': _head(x, w.sub('instance_center_head'), sigmoid=True),
        'instance_offset': _head(x, w.sub('instance_offset_head')),
        'instance_flow': _head(x, w.sub('instance_future_head')) if w.has('instance_future_head.0.weight') else None,
    }
    return {k: (None if v is None else v.view(b, s, *v.shape[1:])) for k, v in out.items()}


# ---------------------------------------------------------------------------------------------
# whole hot path
# ---------------------------------------------------------------------------------------------
def pool_lifted(lifted, geometry, resolution, start, dimension, exact=False):
    """`projection_to_birds_eye_view` on the encoder's native (F, n, C, D, h, w) layout -> (F, C, X, Y).
    fiery/models/fiery.py:221-273.  exact=True: float64 sums instead of the reference's fp32 prefix-sum trick."""
    frames = []
    pool = lift_splat.voxel_pool_exact if exact else lift_splat.voxel_pool_reference
    for f in range(lifted.shape[0]):
        pts = lift_splat.lifted_to_points(lifted[f].numpy())
        frames.append(torch.from_numpy(pool(pts, geometry[f].reshape(-1, 3), resolution, start, dimension)))
    return torch.stack(frames)


def bev_hot_path_exact(sd, cfg, lifted, intrinsics, extrinsics, future_egomotion, noise=None):
    """The same network evaluated in float64 (weights and activations cast up, exact pooling sums; the voxel
    indices still come from the fp32 geometry, as they must).  This is the value the reference's fp32 arithmetic
    approximates; tests use it to measure the reference's own rounding noise and to bound ours."""
    sd64 = {k: (v.double() if v.is_floating_point() and k != 'frustum' else v) for k, v in sd.items()}
    return bev_hot_path(sd64, cfg, lifted, intrinsics, extrinsics, future_egomotion.double(),
                        None if noise is None else noise.double(), exact=True)


def bev_hot_path(sd, cfg, lifted, intrinsics, extrinsics, future_egomotion, noise=None, exact=False):
    """`Fiery.forward` from the lifted features onward (eval mode, no future labels).
    fiery/models/fiery.py:130-191, 275-286, 288-339.

    lifted: (B, S, n, C, D, h, w) = encoder output per frame and camera; intrinsics (B,S,n,3,3),
    extrinsics (B,S,n,4,4), future_egomotion (B,S,6) with S >= receptive field.
    """
    w = Weights(sd)
    rf = cfg.TIME_RECEPTIVE_FIELD
    n_future = cfg.N_FUTURE_FRAMES
    if cfg.MODEL.SUBSAMPLE:
        rf, n_future = 3, 5
    latent = cfg.MODEL.DISTRIBUTION.LATENT_DIM
    resolution, start, dimension = lift_splat.bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
    extent = (cfg.LIFT.X_BOUND[1], cfg.LIFT.Y_BOUND[1])
    lifted = lifted[:, :rf]
    intrinsics, extrinsics = intrinsics[:, :rf], extrinsics[:, :rf]
    ego = future_egomotion[:, :rf].contiguous()
    b, s = lifted.shape[:2]

    frustum = sd['frustum'].numpy()
    geometry = lift_splat.get_geometry(frustum, intrinsics.reshape(b * s, -1, 3, 3).numpy(),
                                       extrinsics.reshape(b * s, -1, 4, 4).numpy())
    x = pool_lifted(lifted.reshape(b * s, *lifted.shape[2:]), geometry, resolution, start, dimension, exact)
    x = x.view(b, s, *x.shape[1:])
    x = cumulative_warp_features(x.clone(), ego, 'bilinear', extent)
    if cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE:
        h, wd = x.shape[-2:]
        spatial = ego.view(b, s, 6, 1, 1).expand(b, s, 6, h, wd)
        spatial = torch.cat([torch.zeros_like(spatial[:, :1]), spatial[:, :(rf - 1)]], dim=1)
        x = torch.cat([x, spatial], dim=-3)
    if cfg.MODEL.TEMPORAL_MODEL.NAME == 'identity':
        states = x[:, (rf - 1):]
    else:
        states = temporal_model(x, w.sub('temporal_model'), rf, cfg.MODEL.TEMPORAL_MODEL.INBETWEEN_LAYERS)
    output = {}
    if n_future > 0:
        present = states[:, :1].contiguous()
        hidden = present[:, 0]
        bb, _, _, h, wd = present.shape
        if cfg.PROBABILISTIC.ENABLED:
            mu, log_sigma = distribution(present, w.sub('present_distribution'), latent,
                                         cfg.MODEL.DISTRIBUTION.MIN_LOG_SIGMA, cfg.MODEL.DISTRIBUTION.MAX_LOG_SIGMA)
            if noise is None:
                noise = torch.zeros_like(mu)
            sample = (mu + torch.exp(log_sigma) * noise).view(bb, 1, latent, 1, 1).expand(bb, 1, latent, h, wd)
            output.update(present_mu=mu, present_log_sigma=log_sigma, future_mu=None, future_log_sigma=None)
            fut_in = sample.expand(-1, n_future, -1, -1, -1)
        else:
            fut_in = hidden.new_zeros(bb, n_future, latent, h, wd)
        fut = future_prediction(fut_in, hidden, w.sub('future_prediction'),
                                cfg.MODEL.FUTURE_PRED.N_GRU_BLOCKS, cfg.MODEL.FUTURE_PRED.N_RES_LAYERS)
        states_out = torch.cat([present, fut], dim=1)
    else:
        states_out = states[:, -1:]
    output.update(decoder(states_out, w.sub('decoder')))
    return output
