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
