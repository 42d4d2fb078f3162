"""Per-frame instance segmentation on the HIP library: the step that follows `Fiery.forward` in evaluation.

Mirrors the reference's `get_instance_segmentation_and_centers` (reference: fiery/utils/instance.py:116-144) - same
arguments, same return values - and adds the batched form `predict_instance_segmentation_and_trajectories` calls it
in (one frame at a time in a Python loop there, :283-291; all frames in one launch here).  The temporal matching
that follows (`make_instance_id_temporally_consistent`, Hungarian assignment on the host, :172-269) is not part of it.
"""
import torch

from . import native


def instance_segmentation_frames(center_predictions, offset_predictions, foreground_mask, conf_threshold=0.1,
                                 nms_kernel_size=3, max_n_instance_centers=100, lib=None):
    """center (n, H, W) / (n, 1, H, W), offset (n, 2, H, W), foreground (n, H, W) bool ->
    (instance ids (n, H, W) int64, centres (n, max, 2) int64 padded with -1, number of centres (n,) int64)."""
    if nms_kernel_size != 3:
        raise ValueError('the HIP kernel implements the reference\'s default 3x3 non-maximum suppression only')
    lib = lib or native.get()
    n = offset_predictions.shape[0]
    h, w = offset_predictions.shape[-2:]
    center = center_predictions.reshape(n, h, w).float().contiguous()
    offset = offset_predictions.reshape(n, 2, h, w).float().contiguous()
    fg = foreground_mask.reshape(n, h, w).to(torch.uint8).contiguous()
    seg, centers, count = lib.instance_segmentation(center, offset, fg, float(conf_threshold), int(max_n_instance_centers))
    return seg.long(), centers.long(), count.long()


def get_instance_segmentation_and_centers(center_predictions, offset_predictions, foreground_mask, conf_threshold=0.1,
                                          nms_kernel_size=3, max_n_instance_centers=100, lib=None):
    """One frame, the reference's signature (instance.py:116-144): -> (instance ids (1, H, W) int64, centres (n, 2))."""
    h, w = center_predictions.shape[-2:]
    seg, centers, count = instance_segmentation_frames(center_predictions.reshape(1, h, w), offset_predictions.reshape(1, 2, h, w),
                                                       foreground_mask.reshape(1, h, w), conf_threshold, nms_kernel_size,
                                                       max_n_instance_centers, lib)
    k = int(count[0])
    if k == 0:
        return seg, torch.zeros((0, 2), device=seg.device)
    return seg, centers[0, :k]
