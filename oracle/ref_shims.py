"""Import the reference's own Python modules from /root/reference on the CPU.

TEST INFRASTRUCTURE.  Works only in the build container (the GPU box has no /root/reference);
used by `tests/golden/make_golden.py` to produce the committed fixtures and by the
`needs_reference` tests that pin `oracle/` against the reference code itself.

The reference imports four third-party packages that are not installed offline
(SURVEY.md Appendix A).  They are replaced in `sys.modules` by minimal stand-ins *before* the
reference is imported; the reference's files are used unmodified and never copied.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('FIERY_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'fiery'))


def _install_shims():
    """The stand-ins for the two networks come from `oracle/third_party.py`, a restatement written independently of the
    product's (`fiery_amd/backbone.py`, `fiery_amd/modules.py`): checker and product do not share a definition."""
    import torch.nn as nn
    from oracle import third_party

    if 'pyquaternion' not in sys.modules:
        mod = types.ModuleType('pyquaternion')
        mod.Quaternion = object           # imported at fiery/utils/geometry.py:5, used only by dataset code
        sys.modules['pyquaternion'] = mod

    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        transforms = types.ModuleType('torchvision.transforms')

        class Normalize(nn.Module):   # base class of NormalizeInverse, fiery/utils/network.py:33
            def __init__(self, mean, std):
                super().__init__()
                self.mean, self.std = mean, std

        transforms.Normalize = Normalize
        models = types.ModuleType('torchvision.models')
        resnet = types.ModuleType('torchvision.models.resnet')
        resnet.resnet18 = third_party.resnet18       # fiery/models/decoder.py:10-17 only touches bn1, relu, layer1..3
        models.resnet = resnet
        tv.transforms, tv.models = transforms, models
        sys.modules.update({'torchvision': tv, 'torchvision.transforms': transforms,
                            'torchvision.models': models, 'torchvision.models.resnet': resnet})

    if 'pytorch_lightning' not in sys.modules:
        # fiery/metrics.py:4-6 needs the Metric base class (state registration only) and two small functions; their
        # semantics are pytorch-lightning 1.1's documented ones: per-class tp / fp / tn / fn / support of class-index
        # maps, and a none / mean / sum reduction
        import torch
        pl = types.ModuleType('pytorch_lightning')
        metrics = types.ModuleType('pytorch_lightning.metrics')
        metric_mod = types.ModuleType('pytorch_lightning.metrics.metric')
        functional = types.ModuleType('pytorch_lightning.metrics.functional')
        classification = types.ModuleType('pytorch_lightning.metrics.functional.classification')
        reduction_mod = types.ModuleType('pytorch_lightning.metrics.functional.reduction')

        class Metric(nn.Module):
            def __init__(self, compute_on_step=False):
                super().__init__()
                self._defaults = {}

            def add_state(self, name, default, dist_reduce_fx=None):
                self._defaults[name] = default.clone()
                setattr(self, name, default.clone())

            def forward(self, *args):
                self.update(*args)

            def reset(self):
                for name, default in self._defaults.items():
                    setattr(self, name, default.clone())

        def stat_scores_multiple_classes(pred, target, num_classes):
            pred, target = pred.reshape(-1), target.reshape(-1)
            tps = torch.stack([((pred == c) & (target == c)).sum() for c in range(num_classes)]).float()
            fps = torch.stack([((pred == c) & (target != c)).sum() for c in range(num_classes)]).float()
            tns = torch.stack([((pred != c) & (target != c)).sum() for c in range(num_classes)]).float()
            fns = torch.stack([((pred != c) & (target == c)).sum() for c in range(num_classes)]).float()
            sups = torch.stack([(target == c).sum() for c in range(num_classes)]).float()
            return tps, fps, tns, fns, sups

        def reduce(to_reduce, reduction):
            if reduction == 'elementwise_mean':
                return torch.mean(to_reduce)
            if reduction == 'none':
                return to_reduce
            if reduction == 'sum':
                return torch.sum(to_reduce)
            raise ValueError('Reduction parameter unknown.')

        metric_mod.Metric = Metric
        classification.stat_scores_multiple_classes = stat_scores_multiple_classes
        reduction_mod.reduce = reduce
        sys.modules.update({'pytorch_lightning': pl, 'pytorch_lightning.metrics': metrics,
                            'pytorch_lightning.metrics.metric': metric_mod, 'pytorch_lightning.metrics.functional': functional,
                            'pytorch_lightning.metrics.functional.classification': classification,
                            'pytorch_lightning.metrics.functional.reduction': reduction_mod})

    if 'efficientnet_pytorch' not in sys.modules:
        eff = types.ModuleType('efficientnet_pytorch')
        eff.EfficientNet = third_party.EfficientNet
        sys.modules['efficientnet_pytorch'] = eff


_REF = None


def load_reference():
    """Returns a namespace with the reference modules (`fiery_model`, `geometry`, `temporal`, ...)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    ns = types.SimpleNamespace()
    ns.geometry = importlib.import_module('fiery.utils.geometry')
    ns.network = importlib.import_module('fiery.utils.network')
    ns.convolutions = importlib.import_module('fiery.layers.convolutions')
    ns.temporal = importlib.import_module('fiery.layers.temporal')
    ns.temporal_model = importlib.import_module('fiery.models.temporal_model')
    ns.future_prediction = importlib.import_module('fiery.models.future_prediction')
    ns.distributions = importlib.import_module('fiery.models.distributions')
    ns.decoder = importlib.import_module('fiery.models.decoder')
    ns.encoder = importlib.import_module('fiery.models.encoder')
    ns.instance = importlib.import_module('fiery.utils.instance')
    ns.metrics = importlib.import_module('fiery.metrics')
    ns.fiery_model = importlib.import_module('fiery.models.fiery')
    ns.Fiery = ns.fiery_model.Fiery
    _REF = ns
    return ns
