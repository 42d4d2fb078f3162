"""Label warping of the training side - the caller that sits before the path in `trainer.py`
(reference: fiery/trainer.py:133-191 `prepare_future_labels`, fiery/utils/geometry.py:256-280
`cumulative_warp_features_reverse`).

Future labels are rendered in each future frame's own ego frame; before they can supervise the prediction (and feed
the future distribution) they are brought into the present frame by the accumulated inverse ego-motion, with
nearest-neighbour sampling so that class and instance ids stay ids.  On the reference that is, per label tensor and
future frame, ~40 small ATen launches for the pose algebra plus affine_grid + grid_sample; here `fiery_warp_params_reverse`
produces every frame's transform in one launch and `fiery_bev_warp_nearest_nchw` resamples every frame of a label tensor
in one launch.
"""
import torch

from . import native


def cumulative_warp_features_reverse(x, flow, mode='nearest', spatial_extent=None, lib=None):
    """x (b, t, c, h, w), flow (b, t, 6) -> x with frame i sampled through inverse(flow[0]) @ ... @ inverse(flow[i-1])
    (frame 0 unchanged).  mode must be 'nearest' (the only one the reference's callers use, trainer.py:146-181)."""
    if mode != 'nearest':
        raise NotImplementedError("cumulative_warp_features_reverse: only mode='nearest' (the label path) is built")
    lib = lib or native.get()
    b, t, c, h, w = x.shape
    theta = lib.warp_params_reverse(flow.float().contiguous(), spatial_extent)
    out = lib.bev_warp_nearest(x.float().contiguous().view(b * t, c, h, w), theta.view(b * t, 6))
    return out.view(b, t, c, h, w).to(x.dtype)


def prepare_future_labels(batch, receptive_field, spatial_extent, instance_flow_enabled=True, lib=None):
    """`FieryTrainer.prepare_future_labels` (fiery/trainer.py:133-191): dataset batch -> (labels dict, future distribution
    inputs (b, 1 + n_future, 6, h, w)), everything warped into the present frame."""
    start = receptive_field - 1
    ego = batch['future_egomotion'][:, start:]
    warp = lambda t: cumulative_warp_features_reverse(t[:, start:], ego, mode='nearest', spatial_extent=spatial_extent, lib=lib)
    labels = {}
    future_distribution_inputs = []
    segmentation = warp(batch['segmentation'].float()).long().contiguous()
    labels['segmentation'] = segmentation
    future_distribution_inputs.append(segmentation)
    labels['instance'] = warp(batch['instance'].float().unsqueeze(2)).long().contiguous()[:, :, 0]
    labels['centerness'] = warp(batch['centerness']).contiguous()
    labels['offset'] = warp(batch['offset']).contiguous()
    future_distribution_inputs += [labels['centerness'], labels['offset']]
    if instance_flow_enabled:
        labels['flow'] = warp(batch['flow']).contiguous()
        future_distribution_inputs.append(labels['flow'])
    return labels, torch.cat(future_distribution_inputs, dim=2)


def _pose_vec2mat(vec):
    """(n, 6) (tx, ty, tz, rx, ry, rz) -> (n, 4, 4), R = Rx . Ry . Rz (fiery/utils/geometry.py:109-157)."""
    x, y, z = vec[:, 3], vec[:, 4], vec[:, 5]
    zeros, ones = torch.zeros_like(x), torch.ones_like(x)
    rot_z = torch.stack([torch.cos(z), -torch.sin(z), zeros, torch.sin(z), torch.cos(z), zeros, zeros, zeros, ones], dim=1).view(-1, 3, 3)
    rot_y = torch.stack([torch.cos(y), zeros, torch.sin(y), zeros, ones, zeros, -torch.sin(y), zeros, torch.cos(y)], dim=1).view(-1, 3, 3)
    rot_x = torch.stack([ones, zeros, zeros, zeros, torch.cos(x), -torch.sin(x), zeros, torch.sin(x), torch.cos(x)], dim=1).view(-1, 3, 3)
    top = torch.cat([rot_x.bmm(rot_y).bmm(rot_z), vec[:, :3].unsqueeze(-1)], dim=2)
    mat = torch.cat([top, top.new_zeros(top.shape[0], 1, 4)], dim=1)
    mat[:, 3, 3] = 1.0
    return mat


def _mat2pose_vec(matrix):
    """(n, 4, 4) -> (n, 6) (fiery/utils/geometry.py:82-106)."""
    rotx = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    cosy = torch.sqrt(matrix[..., 1, 2] ** 2 + matrix[..., 2, 2] ** 2)
    roty = torch.atan2(matrix[..., 0, 2], cosy)
    rotz = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rotx, roty, rotz), dim=-1)), dim=-1)


def convert_instance_mask_to_center_and_offset_label(instance_img, future_egomotion, num_instances, ignore_index=255,
                                                     subtract_egomotion=True, sigma=3, spatial_extent=None, lib=None, device=None):
    """The dataset's instance labels (fiery/utils/instance.py:12-77, called by fiery/data.py for every sample): instance-id maps
    (seq_len, h, w) + ego-motion (seq_len, 6) -> centerness (seq_len, 1, h, w), offset (seq_len, 2, h, w) and future displacement
    (seq_len, 2, h, w), `ignore_index` where undefined.  Same arguments and results as the reference; the per-instance, per-frame
    Python loop is two kernel launches (`fiery_instance_labels`), the id maps are resampled into the previous frame by
    `fiery_bev_warp_nearest_nchw`.  The poses - a handful of 4 x 4 matrices - are inverted with the reference's own operators
    on the host.  Tensors may live on the host or on the GPU; host inputs are moved to `device` (default: the current HIP device -
    there is no CPU path) and the results brought back to where the input was.  In a DataLoader worker this needs the
    'spawn' start method (a forked worker cannot use the parent's HIP context: the call raises and says so)."""
    if not subtract_egomotion:
        raise ValueError('convert_instance_mask_to_center_and_offset_label: the reference only defines the warped instance maps '
                         'with subtract_egomotion=True (fiery/utils/instance.py:22-31)')
    native.require_usable_gpu_process('convert_instance_mask_to_center_and_offset_label')
    lib = lib or native.get()
    seq_len, h, w = instance_img.shape
    where = instance_img.device
    device = torch.device(device) if device is not None else (where if where.type == 'cuda' else torch.device('cuda'))
    ids = instance_img.to(device=device, dtype=torch.int32).contiguous()
    ego = future_egomotion.detach().float().cpu()
    inverse = _mat2pose_vec(torch.inverse(_pose_vec2mat(ego)))                   # instance.py:22-23
    angle, tx, ty = inverse[:, 5], inverse[:, 0] / spatial_extent[0], inverse[:, 1] / spatial_extent[1]
    cos, sin = torch.cos(angle), torch.sin(angle)
    theta = torch.stack([cos, -sin, ty, sin, cos, -tx], dim=-1)                   # warp_features, geometry.py:181-222
    warped = torch.zeros_like(ids)
    if seq_len > 1:
        frames = ids[1:].float().view(seq_len - 1, 1, h, w)
        warped[1:] = lib.bev_warp_nearest(frames, theta[:seq_len - 1].to(device).contiguous()).view(seq_len - 1, h, w).to(torch.int32)
    center, offset, flow = lib.instance_labels(ids, warped, int(num_instances), sigma, ignore_index)
    return center.to(where), offset.to(where), flow.to(where)
