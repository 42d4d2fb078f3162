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

        class _Recorder:
            """What `self.logger.experiment` is asked to do, kept for the test to look at."""
            def __init__(self):
                self.scalars, self.videos = [], []

            def add_scalar(self, key, value, global_step=None):
                self.scalars.append((key, float(value), global_step))

            def add_video(self, name, video, global_step=None, fps=None):
                self.videos.append((name, global_step))

        class LightningModule(nn.Module):
            """pytorch-lightning 1.1's LightningModule as far as fiery/trainer.py and evaluate.py use it: an nn.Module with a
            settable `hparams`, `log`, a `logger.experiment`, and `load_from_checkpoint` (a `torch.save`d dict with
            'state_dict' and 'hyper_parameters', the two checkpoint keys the class reads)."""
            def __init__(self):
                super().__init__()
                self.logged = {}
                self.logger = types.SimpleNamespace(experiment=_Recorder())

            def log(self, key, value, **kwargs):
                self.logged[key] = value

            @classmethod
            def load_from_checkpoint(cls, checkpoint_path, strict=True, **kwargs):
                ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
                module = cls(ckpt['hyper_parameters'])
                module.load_state_dict(ckpt['state_dict'], strict=strict)
                return module

        pl.LightningModule = LightningModule
        metric_mod.Metric = Metric
        classification.stat_scores_multiple_classes = stat_scores_multiple_classes
        reduction_mod.reduce = reduce
        sys.modules.update({'pytorch_lightning': pl, 'pytorch_lightning.metrics': metrics,
                            'pytorch_lightning.metrics.metric': metric_mod, 'pytorch_lightning.metrics.functional': functional,
                            'pytorch_lightning.metrics.functional.classification': classification,
                            'pytorch_lightning.metrics.functional.reduction': reduction_mod})

    if 'fvcore' not in sys.modules:
        # fiery/config.py:2 builds its defaults on fvcore's CfgNode (yacs + `_BASE_` files): configuration plumbing, restated
        # here as far as config.py:32-148 and trainer.py:21 use it
        import copy
        import yaml
        fv = types.ModuleType('fvcore')
        common = types.ModuleType('fvcore.common')
        config = types.ModuleType('fvcore.common.config')

        class CfgNode(dict):
            def __init__(self, init_dict=None):
                super().__init__()
                for k, v in (init_dict or {}).items():
                    self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v
                self.__dict__['_frozen'] = False

            def __getattr__(self, name):
                try:
                    return self[name]
                except KeyError:
                    raise AttributeError(name)

            def __setattr__(self, name, value):
                if self.__dict__.get('_frozen'):
                    raise AttributeError(f'attempted to modify the frozen config key {name}')
                self[name] = value

            def clone(self):
                return copy.deepcopy(self)

            def __deepcopy__(self, memo):
                new = type(self)()
                for k, v in self.items():
                    dict.__setitem__(new, k, copy.deepcopy(v, memo))
                return new

            def freeze(self):
                self._set_frozen(True)

            def defrost(self):
                self._set_frozen(False)

            def _set_frozen(self, flag):
                self.__dict__['_frozen'] = flag
                for v in self.values():
                    if isinstance(v, CfgNode):
                        v._set_frozen(flag)

            def merge_from_other_cfg(self, other):
                for k, v in other.items():
                    if k not in self:
                        raise KeyError(f'non-existent config key: {k}')
                    if isinstance(v, dict) and isinstance(self[k], CfgNode):
                        self[k].merge_from_other_cfg(v if isinstance(v, CfgNode) else CfgNode(v))
                    else:
                        cur = self[k]
                        if isinstance(cur, tuple) and isinstance(v, list):
                            v = tuple(v)
                        elif isinstance(cur, list) and isinstance(v, tuple):
                            v = list(v)
                        dict.__setitem__(self, k, copy.deepcopy(v))

            def merge_from_file(self, path):
                def load(pth):
                    with open(pth) as f:
                        raw = yaml.safe_load(f) or {}
                    base = raw.pop('_BASE_', None)
                    if base is None:
                        return raw
                    if not os.path.isabs(base):
                        base = os.path.join(os.path.dirname(pth), base)
                    merged = load(base)

                    def overlay(dst, src):
                        for k, v in src.items():
                            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                                overlay(dst[k], v)
                            else:
                                dst[k] = v
                    overlay(merged, raw)
                    return merged
                self.merge_from_other_cfg(CfgNode(load(path)))

            def merge_from_list(self, opts):
                import ast
                for key, value in zip(opts[0::2], opts[1::2]):
                    node = self
                    parts = key.split('.')
                    for part in parts[:-1]:
                        node = node[part]
                    if isinstance(value, str):
                        try:
                            value = ast.literal_eval(value)
                        except (ValueError, SyntaxError):
                            pass
                    node.merge_from_other_cfg(CfgNode({parts[-1]: value}))

        config.CfgNode = CfgNode
        fv.common, common.config = common, config
        sys.modules.update({'fvcore': fv, 'fvcore.common': common, 'fvcore.common.config': config})

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


def load_reference_callers(batches=None):
    """The reference's caller code, unmodified: `fiery.config`, `fiery.losses`, `fiery.trainer` (TrainingModule) and, when
    `batches` is given, `evaluate` (repository root) with `fiery.data.prepare_dataloaders` standing in for the nuScenes /
    Lyft loaders (absent packages, absent datasets): it returns `batches` as the validation loader.  Returns a namespace."""
    import importlib
    ns = load_reference()
    ns.config = importlib.import_module('fiery.config')
    ns.losses = importlib.import_module('fiery.losses')
    ns.trainer = importlib.import_module('fiery.trainer')
    if batches is not None:
        data = types.ModuleType('fiery.data')
        data.prepare_dataloaders = lambda cfg: (None, batches)
        sys.modules['fiery.data'] = data
        ns.evaluate_path = os.path.join(REFERENCE_ROOT, 'evaluate.py')
    return ns


class ScipyQuaternion:
    """Stand-in for `pyquaternion.Quaternion` (absent offline) on top of scipy's `Rotation` - the three members the
    reference's dataset code touches (fiery/data.py:173-200, fiery/utils/geometry.py:63), written independently of
    `fiery_amd/poses.py`, which restates pyquaternion's own formulas.  (w, x, y, z) order, like the package."""

    def __init__(self, array=None, scalar=None, vector=None):
        import numpy as np
        if array is None:
            array = [scalar] + list(vector)
        self.q = np.asarray(array, dtype=np.float64)

    def _rotation(self):
        from scipy.spatial.transform import Rotation
        w, x, y, z = self.q
        return Rotation.from_quat([x, y, z, w])              # scipy: scalar last; normalises

    @property
    def rotation_matrix(self):
        return self._rotation().as_matrix()

    @property
    def inverse(self):
        import numpy as np
        w, x, y, z = self.q
        return ScipyQuaternion(np.array([w, -x, -y, -z]) / float(np.dot(self.q, self.q)))

    @property
    def yaw_pitch_roll(self):
        # the package documents R = R_x(roll) R_y(pitch) R_z(yaw): scipy's extrinsic z-y-x sequence
        return tuple(self._rotation().as_euler('zyx'))


_REF_DATA = None


def load_reference_data():
    """`fiery.data` of the reference, importable offline: cv2, the two dataset SDKs and torchvision's image transforms are
    replaced by minimal stand-ins (none of them is reached by `get_input_data` / `get_future_egomotion` except the
    transforms, whose documented semantics are two lines), `Quaternion` by `ScipyQuaternion`."""
    global _REF_DATA
    if _REF_DATA is not None:
        return _REF_DATA
    ns = load_reference()
    import importlib
    import numpy as np
    import torch
    for name in ('cv2', 'nuscenes', 'nuscenes.nuscenes', 'nuscenes.utils', 'nuscenes.utils.splits', 'nuscenes.utils.data_classes',
                 'lyft_dataset_sdk', 'lyft_dataset_sdk.lyftdataset'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['nuscenes.nuscenes'].NuScenes = type('NuScenes', (), {})
    sys.modules['nuscenes.utils.splits'].create_splits_scenes = lambda: {}
    sys.modules['nuscenes.utils.data_classes'].Box = type('Box', (), {})
    sys.modules['lyft_dataset_sdk.lyftdataset'].LyftDataset = type('LyftDataset', (), {})
    transforms = sys.modules['torchvision.transforms']

    class ToTensor:                       # PIL (H, W, C) uint8 -> (C, H, W) float in [0, 1]
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)

    class Compose:
        def __init__(self, steps):
            self.steps = steps

        def __call__(self, x):
            for step in self.steps:
                x = step(x)
            return x

    def normalize_call(self, x):          # torchvision.transforms.Normalize: (x - mean) / std per channel
        mean = torch.as_tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
        return (x - mean) / std

    transforms.ToTensor, transforms.Compose = ToTensor, Compose
    transforms.Normalize.forward = normalize_call
    sys.modules['torchvision'].transforms = transforms
    data = importlib.import_module('fiery.data')
    data.Quaternion = ScipyQuaternion
    ns.geometry.Quaternion = ScipyQuaternion
    ns.data = data
    _REF_DATA = ns
    return ns
