"""Evaluation metrics of the step after the path (reference: fiery/metrics.py:9-255, used by trainer.py:57-58,119-126 and
evaluate.py:41-77): semantic IoU and the video panoptic quality.

The reference derives both from `pytorch_lightning.metrics.Metric` (absent offline, and only used for its state handling
and the multi-process sum); here they are plain classes with the same constructor arguments, `update` / `compute` / `reset`
and `__call__`, whose states are tensors that live where the inputs live.  `sync(group)` sums the states over a
`torch.distributed` group (what `dist_reduce_fx='sum'` does in the reference under DDP).
"""
import torch


class _SummedState:
    def __init__(self):
        self._names = []

    def _add_state(self, name, n):
        self._names.append(name)
        setattr(self, name, torch.zeros(n))

    def reset(self):
        for name in self._names:
            setattr(self, name, torch.zeros_like(getattr(self, name)))

    def _to(self, device):
        for name in self._names:
            if getattr(self, name).device != device:
                setattr(self, name, getattr(self, name).to(device))

    def sync(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            for name in self._names:
                dist.all_reduce(getattr(self, name), group=group)

    def __call__(self, *args):
        self.update(*args)
        return None


class IntersectionOverUnion(_SummedState):
    """fiery/metrics.py:9-68.  prediction / target: integer class maps of any (equal) shape."""

    def __init__(self, n_classes, ignore_index=None, absent_score=0.0, reduction='none'):
        super().__init__()
        self.n_classes, self.ignore_index, self.absent_score, self.reduction = n_classes, ignore_index, absent_score, reduction
        for name in ('true_positive', 'false_positive', 'false_negative', 'support'):
            self._add_state(name, n_classes)

    def update(self, prediction, target):
        self._to(prediction.device)
        n = self.n_classes
        pred, tgt = prediction.reshape(-1).long(), target.reshape(-1).long()
        # one confusion matrix instead of a pass per class: row = target class, column = predicted class
        valid = (pred >= 0) & (pred < n) & (tgt >= 0) & (tgt < n)
        conf = torch.bincount(tgt[valid] * n + pred[valid], minlength=n * n).view(n, n).float()
        tp = conf.diag()
        pred_count = torch.bincount(pred[(pred >= 0) & (pred < n)], minlength=n).float()
        tgt_count = torch.bincount(tgt[(tgt >= 0) & (tgt < n)], minlength=n).float()
        self.true_positive += tp
        self.false_positive += pred_count - tp
        self.false_negative += tgt_count - tp
        self.support += tgt_count

    def compute(self):
        tp, fp, fn, sup = self.true_positive, self.false_positive, self.false_negative, self.support
        absent = (sup + tp + fp) == 0
        scores = torch.where(absent, torch.full_like(tp, self.absent_score), tp / (tp + fp + fn).clamp(min=1e-30))
        if self.ignore_index is not None and 0 <= self.ignore_index < self.n_classes:
            keep = torch.ones(self.n_classes, dtype=torch.bool, device=scores.device)
            keep[self.ignore_index] = False
            scores = scores[keep]
        if self.reduction == 'elementwise_mean':
            return scores.mean()
        if self.reduction == 'sum':
            return scores.sum()
        return scores


class PanopticMetric(_SummedState):
    """fiery/metrics.py:71-255: panoptic quality over a video; an instance whose matched prediction changes id from one
    frame to the next counts as a miss and a false alarm (`temporally_consistent`)."""

    def __init__(self, n_classes, temporally_consistent=True, vehicles_id=1):
        super().__init__()
        self.n_classes, self.temporally_consistent, self.vehicles_id = n_classes, temporally_consistent, vehicles_id
        self.keys = ['iou', 'true_positive', 'false_positive', 'false_negative']
        for name in self.keys:
            self._add_state(name, n_classes)

    def update(self, pred_instance, gt_instance):
        """pred_instance, gt_instance: (b, s, h, w) instance ids, 0 = background."""
        self._to(gt_instance.device)
        assert gt_instance.min() == 0, 'ID 0 of gt_instance must be background'
        pred_segmentation, gt_segmentation = (pred_instance > 0).long(), (gt_instance > 0).long()
        for b in range(gt_instance.shape[0]):
            unique_id_mapping = {}
            for t in range(gt_instance.shape[1]):
                result = self.panoptic_metrics(pred_segmentation[b, t].detach(), pred_instance[b, t].detach(),
                                               gt_segmentation[b, t], gt_instance[b, t], unique_id_mapping)
                for key in self.keys:
                    setattr(self, key, getattr(self, key) + result[key])

    def compute(self):
        denominator = torch.maximum(self.true_positive + self.false_positive / 2 + self.false_negative / 2,
                                    torch.ones_like(self.true_positive))
        return {'pq': self.iou / denominator,
                'sq': self.iou / torch.maximum(self.true_positive, torch.ones_like(self.true_positive)),
                'rq': self.true_positive / denominator,
                'denominator': self.true_positive + self.false_positive / 2 + self.false_negative / 2}

    def panoptic_metrics(self, pred_segmentation, pred_instance, gt_segmentation, gt_instance, unique_id_mapping):
        n_classes = self.n_classes
        device = gt_instance.device
        result = {key: torch.zeros(n_classes, dtype=torch.float32, device=device) for key in self.keys}
        assert pred_segmentation.dim() == 2
        assert pred_segmentation.shape == pred_instance.shape == gt_segmentation.shape == gt_instance.shape
        n_instances = int(torch.cat([pred_instance, gt_instance]).max().item())
        n_all_things = n_instances + n_classes
        n_things_and_void = n_all_things + 1
        prediction, pred_to_cls = self.combine_mask(pred_segmentation, pred_instance, n_classes, n_all_things)
        target, target_to_cls = self.combine_mask(gt_segmentation, gt_instance, n_classes, n_all_things)
        # joint histogram of (target segment, predicted segment), void row / column dropped
        conf = torch.bincount((prediction + n_things_and_void * target).long(), minlength=n_things_and_void ** 2)
        conf = conf.view(n_things_and_void, n_things_and_void)[1:, 1:]
        union = conf.sum(0).unsqueeze(0) + conf.sum(1).unsqueeze(1) - conf
        iou = torch.where(union > 0, (conf.float() + 1e-9) / (union.float() + 1e-9), torch.zeros_like(union).float())
        mapping = (iou > 0.5).nonzero(as_tuple=False)                        # (target segment, predicted segment) pairs
        mapping = mapping[pred_to_cls[mapping[:, 1]] == target_to_cls[mapping[:, 0]]]
        tp_mask = torch.zeros_like(conf, dtype=torch.bool)
        tp_mask[mapping[:, 0], mapping[:, 1]] = True
        # the id bookkeeping is sequential by nature (a dict carried from frame to frame): a few dozen pairs per frame
        pairs = mapping.cpu().tolist()
        cls_of_pred, cls_of_target = pred_to_cls.cpu().tolist(), target_to_cls.cpu().tolist()
        for target_id, pred_id in pairs:
            cls_id = cls_of_pred[pred_id]
            if self.temporally_consistent and cls_id == self.vehicles_id:
                if target_id in unique_id_mapping and unique_id_mapping[target_id] != pred_id:
                    result['false_negative'][cls_of_target[target_id]] += 1
                    result['false_positive'][cls_of_pred[pred_id]] += 1
                    unique_id_mapping[target_id] = pred_id
                    continue
            result['true_positive'][cls_id] += 1
            result['iou'][cls_id] += iou[target_id][pred_id]
            unique_id_mapping[target_id] = pred_id
        # instances of either side that found no partner
        thing = torch.arange(n_classes, n_all_things, device=device)
        if len(thing):
            missed = ~tp_mask[thing][:, n_classes:].any(dim=1) & (target_to_cls[thing] != -1)
            result['false_negative'].index_add_(0, target_to_cls[thing][missed].clamp(min=0), torch.ones(int(missed.sum()), device=device))
            spurious = (~tp_mask[n_classes:][:, thing].any(dim=0) & (pred_to_cls[thing] != -1) & (conf[:, thing] > 0).any(dim=0))
            result['false_positive'].index_add_(0, pred_to_cls[thing][spurious].clamp(min=0), torch.ones(int(spurious.sum()), device=device))
        return result

    def combine_mask(self, segmentation, instance, n_classes, n_all_things):
        """Things and stuff in one id map (0 = void, 1 .. n_classes = stuff, then the instances) and the class of every id."""
        instance = instance.reshape(-1)
        instance_mask = instance > 0
        instance = instance - 1 + n_classes
        segmentation = segmentation.clone().reshape(-1)
        segmentation_mask = segmentation < n_classes
        both = instance_mask & segmentation_mask
        instance_id_to_class = -torch.ones(n_all_things, dtype=segmentation.dtype, device=segmentation.device)
        instance_id_to_class[instance[both]] = segmentation[both]
        instance_id_to_class[:n_classes] = torch.arange(n_classes, device=segmentation.device, dtype=segmentation.dtype)
        segmentation[instance_mask] = instance[instance_mask]
        segmentation += 1
        segmentation[~segmentation_mask] = 0
        return segmentation, instance_id_to_class
