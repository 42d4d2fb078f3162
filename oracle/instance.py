"""CPU restatement of the per-frame instance segmentation that follows the hot path in evaluation.
TEST INFRASTRUCTURE (see oracle/__init__.py).

reference: fiery/utils/instance.py:80-144 (`find_instance_centers`, `group_pixels`, `get_instance_segmentation_and_centers`,
`make_instance_seg_consecutive`), called per frame by `predict_instance_segmentation_and_trajectories` (:272-300) from
evaluate.py:62.  The same torch CPU operators in the same order, so integer results are the reference's bit for bit
(pinned in tests/test_oracle_vs_reference.py).
"""
import torch
import torch.nn.functional as F


def find_instance_centers(center_prediction, conf_threshold=0.1, nms_kernel_size=3):
    """(1, H, W) centerness -> (n, 2) int64 coordinates of the local maxima above the threshold, row-major order.
    reference: instance.py:80-92."""
    center_prediction = F.threshold(center_prediction.clone(), threshold=conf_threshold, value=-1)
    pad = (nms_kernel_size - 1) // 2
    pooled = F.max_pool2d(center_prediction, kernel_size=nms_kernel_size, stride=1, padding=pad)
    center_prediction[center_prediction != pooled] = -1
    return torch.nonzero(center_prediction > 0)[:, 1:]


def group_pixels(centers, offset_predictions):
    """Every pixel joins the centre nearest to (pixel + predicted offset); ids start at 1.  reference: instance.py:95-113."""
    width, height = offset_predictions.shape[-2:]
    x_grid = torch.arange(width, dtype=offset_predictions.dtype).view(1, width, 1).repeat(1, 1, height)
    y_grid = torch.arange(height, dtype=offset_predictions.dtype).view(1, 1, height).repeat(1, width, 1)
    pixel_grid = torch.cat((x_grid, y_grid), dim=0)
    center_locations = (pixel_grid + offset_predictions).view(2, width * height, 1).permute(2, 1, 0)
    centers = centers.view(-1, 1, 2)
    distances = torch.norm(centers - center_locations, dim=-1)
    return torch.argmin(distances, dim=0).reshape(1, width, height) + 1


def make_instance_seg_consecutive(instance_seg):
    """Present ids, sorted, renumbered 0, 1, 2, ... (0 only stays the background if a background pixel exists).
    reference: instance.py:164-169 with update_instance_ids :147-161."""
    unique_ids = torch.unique(instance_seg)
    lut = torch.arange(int(unique_ids.max()) + 1)
    lut[unique_ids] = torch.arange(len(unique_ids))
    return lut[instance_seg].long()


def instance_segmentation_and_centers(center_predictions, offset_predictions, foreground_mask, conf_threshold=0.1,
                                      nms_kernel_size=3, max_n_instance_centers=100):
    """One frame: centerness (H, W), offsets (2, H, W), foreground mask (H, W) -> (instance ids (1, H, W) int64,
    centres (n, 2)).  reference: instance.py:116-144."""
    width, height = center_predictions.shape[-2:]
    center_predictions = center_predictions.reshape(1, width, height)
    offset_predictions = offset_predictions.reshape(2, width, height)
    foreground_mask = foreground_mask.reshape(1, width, height)
    centers = find_instance_centers(center_predictions, conf_threshold, nms_kernel_size)
    if not len(centers):
        return torch.zeros(center_predictions.shape, dtype=torch.int64), torch.zeros((0, 2))
    if len(centers) > max_n_instance_centers:
        centers = centers[:max_n_instance_centers].clone()
    instance_ids = group_pixels(centers, offset_predictions)
    instance_seg = (instance_ids * foreground_mask.float()).long()
    return make_instance_seg_consecutive(instance_seg).long(), centers
