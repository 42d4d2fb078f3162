"""Per-frame instance segmentation on the HIP library: the step that follows `Fiery.forward` in evaluation.

Mirrors the reference's `get_instance_segmentation_and_centers` (reference: fiery/utils/instance.py:116-144) - same
arguments, same return values - and the entry point evaluation uses, `predict_instance_segmentation_and_trajectories`
(:272-330; one frame at a time in a Python loop there, all frames in one launch here).  The temporal matching that
follows (`make_instance_id_temporally_consistent`, :172-269) is host logic on small tensors in the reference too: it is
restated here with PyTorch operators and the same scipy Hungarian assignment, no kernel.
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


def _instance_means(ids, values, n_ids):
    """Mean of `values` (2, H, W) over the pixels of every id 0 .. n_ids-1 of `ids` (H, W): (n_ids, 2) float64 and the
    pixel counts - one scatter-add instead of a Python loop over masks."""
    flat = ids.reshape(-1)
    sums = torch.zeros(n_ids, 2, dtype=torch.float64, device=ids.device)
    sums.index_add_(0, flat, values.reshape(2, -1).t().double())
    counts = torch.bincount(flat, minlength=n_ids).double()
    return sums / counts.clamp(min=1).unsqueeze(1), counts


def make_instance_id_temporally_consistent(pred_inst, future_flow, matching_threshold=3.0):
    """pred_inst (1, T, H, W) per-frame ids, future_flow (1, T, 2, H, W) -> (1, T, H, W) ids that follow an instance
    through time: the instances of frame t, moved by the predicted flow, are matched to those of frame t+1 by position
    (Hungarian assignment, matches further apart than the threshold are dropped), unmatched instances of t+1 get fresh
    ids.  Same algorithm and the same `scipy.optimize.linear_sum_assignment` as the reference
    (fiery/utils/instance.py:172-269); the per-instance means are one scatter-add per frame rather than a loop over masks."""
    from scipy.optimize import linear_sum_assignment
    assert pred_inst.shape[0] == 1, 'Assumes batch size = 1'
    _, seq_len, h, w = pred_inst.shape
    device = pred_inst.device
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float, device=device), torch.arange(w, dtype=torch.float, device=device),
                            indexing='ij')
    grid = torch.stack((yy, xx))
    frames = [pred_inst[0, 0]]
    largest = int(frames[0].max().item())
    for t in range(seq_len - 1):
        cur, nxt = frames[-1], pred_inst[0, t + 1]
        n_next = int(nxt.max().item())
        cur_means, cur_counts = _instance_means(cur, grid + future_flow[0, t], int(cur.max().item()) + 1)
        cur_ids = torch.nonzero(cur_counts[1:] > 0).flatten() + 1              # ids present at t, ascending, no background
        if len(cur_ids) == 0 or n_next == 0:
            frames.append(nxt)
            continue
        nxt_means, nxt_counts = _instance_means(nxt, grid, n_next + 1)
        warped = cur_means[cur_ids].float()                                    # where the instances of t should be at t+1
        centres = nxt_means[1:].float()                                        # where the instances of t+1 are (ids 1..n)
        # (per-frame ids are consecutive - the segmentation renumbers them - so every id 1..n owns pixels)
        distances = torch.norm(centres.unsqueeze(0) - warped.unsqueeze(1), dim=-1).cpu().numpy()
        rows, cols = linear_sum_assignment(distances)
        good = distances[rows, cols] < matching_threshold
        new_ids = cur_ids.cpu().numpy()[rows[good]]
        old_ids = cols[good] + 1
        lut = torch.zeros(n_next + 1, dtype=torch.long, device=device)
        lut[torch.as_tensor(old_ids, dtype=torch.long, device=device)] = torch.as_tensor(new_ids, dtype=torch.long, device=device)
        # new instances get fresh ids in the order the reference visits them: it iterates a Python set of numpy integers
        # built exactly like this one (instance.py:256-262) - a set's order is not ascending in general, and the ids
        # handed out follow it, so the same set is built the same way here
        remaining = set(torch.unique(nxt).cpu().numpy()).difference(set(old_ids))
        remaining.remove(0)
        for rid in list(remaining):
            largest += 1
            lut[int(rid)] = largest
        frames.append(lut[nxt])
    return torch.stack(frames).unsqueeze(0)


def predict_instance_segmentation_and_trajectories(output, compute_matched_centers=False, make_consistent=True, vehicles_id=1,
                                                   lib=None):
    """The reference's post-processing entry point (fiery/utils/instance.py:272-330, called from evaluate.py:62): model
    outputs -> (B, T, H, W) instance ids, consistent through time.  All B*T frames are segmented by one HIP launch."""
    seg = output['segmentation'].detach()
    b, t = seg.shape[:2]
    h, w = seg.shape[-2:]
    foreground = torch.argmax(seg, dim=2) == vehicles_id
    ids, _, _ = instance_segmentation_frames(output['instance_center'].detach().reshape(b * t, h, w),
                                             output['instance_offset'].detach().reshape(b * t, 2, h, w),
                                             foreground.reshape(b * t, h, w), lib=lib)
    pred_inst = ids.view(b, t, h, w)
    if make_consistent:
        flow = output.get('instance_flow')
        if flow is None:
            flow = torch.zeros_like(output['instance_offset'])
        consistent = torch.cat([make_instance_id_temporally_consistent(pred_inst[i:i + 1], flow[i:i + 1].detach())
                                for i in range(b)], 0)
    else:
        consistent = pred_inst
    if not compute_matched_centers:
        return consistent
    return consistent, matched_centers(consistent)


def matched_centers(consistent_instance_seg):
    """The visualiser's trajectories (fiery/utils/instance.py:308-328): for every instance present in the first frame,
    the mean pixel position of its mask in every frame where it appears, as an (n_frames_present, 2) numpy array in (x, y)
    order.  One scatter-add per frame instead of a loop over instances x frames."""
    assert consistent_instance_seg.shape[0] == 1
    seg = consistent_instance_seg[0].long()
    t_len, h, w = seg.shape
    device = seg.device
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float, device=device), torch.arange(w, dtype=torch.float, device=device),
                            indexing='ij')
    grid = torch.stack((yy, xx))
    n_ids = int(seg.max().item()) + 1
    per_frame = [_instance_means(seg[t], grid, n_ids) for t in range(t_len)]
    centers = {}
    for instance_id in torch.unique(seg[0])[1:].cpu().numpy():
        rows = [means[int(instance_id)].float() for means, counts in per_frame if counts[int(instance_id)] > 0]
        if rows:
            centers[instance_id] = torch.stack(rows).cpu().numpy()[:, ::-1]
    return centers
