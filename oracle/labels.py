"""TEST INFRASTRUCTURE: CPU restatement of the dataset's instance-label generation.

`convert_instance_mask_to_center_and_offset_label` of fiery/utils/instance.py:12-77 (called by fiery/data.py for every sample),
restated with numpy loops over instances and frames, the pose inversion and the nearest-neighbour resampling of the id maps with
the reference's own torch operators (`oracle.bev_stack`, pinned bitwise).  Pinned against the reference function itself in
tests/test_oracle_vs_reference.py.  Only tests import this module.
"""
import numpy as np
import torch

from oracle import bev_stack


def instance_labels(instance_img, future_egomotion, num_instances, ignore_index=255, sigma=3, spatial_extent=None):
    """instance_img (T, H, W) integer ids, future_egomotion (T, 6) -> centerness (T,1,H,W), offset (T,2,H,W), flow (T,2,H,W)."""
    seq_len, h, w = instance_img.shape
    ids = instance_img.numpy().astype(np.int64)
    center = np.zeros((seq_len, 1, h, w), np.float32)
    offset = np.full((seq_len, 2, h, w), ignore_index, np.float32)
    flow = np.full((seq_len, 2, h, w), ignore_index, np.float32)
    rows, cols = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    # instance.py:22-31: every later frame's id map resampled by the inverse of the ego-motion that led to it
    inverse = bev_stack.matrix_to_pose(torch.inverse(bev_stack.pose_to_matrix(future_egomotion.float())))
    warped = {}
    for t in range(1, seq_len):
        img = instance_img[t].float().view(1, 1, h, w)
        warped[t] = bev_stack.warp_features(img, inverse[t - 1:t], 'nearest', spatial_extent)[0, 0].numpy().astype(np.int64)

    def centre(mask):                                            # .mean().round(): fp32 mean, round half to even
        n = np.float32(mask.sum())
        return (np.rint(np.float32(rows[mask].sum(dtype=np.float64)) / n).astype(np.float32),
                np.rint(np.float32(cols[mask].sum(dtype=np.float64)) / n).astype(np.float32))

    for k in range(1, num_instances + 1):
        previous = None
        for t in range(seq_len):
            mask = ids[t] == k
            if not mask.any():
                previous = None
                continue
            xc, yc = centre(mask)
            off_x, off_y = xc - rows, yc - cols
            g = np.exp(-(off_x * off_x + off_y * off_y) / np.float32(sigma ** 2)).astype(np.float32)
            center[t, 0] = np.maximum(center[t, 0], g)
            offset[t, 0][mask] = off_x[mask]
            offset[t, 1][mask] = off_y[mask]
            if previous is not None:
                moved = warped[t] == k
                if moved.any():
                    wx, wy = centre(moved)
                    flow[t - 1, 0][previous[2]] = wx - previous[0]
                    flow[t - 1, 1][previous[2]] = wy - previous[1]
            previous = (xc, yc, mask)
    return torch.from_numpy(center), torch.from_numpy(offset), torch.from_numpy(flow)
