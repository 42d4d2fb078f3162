"""Configuration surface of the BEV hot path.

Mirrors the *behaviour* of the reference's yacs/fvcore-based loader
(reference: fiery/config.py:32-123 defaults, :126-133 parser, :136-149 get_cfg)
without depending on fvcore/yacs (absent offline):

* nested attribute-access nodes (`cfg.LIFT.X_BOUND`),
* YAML overlays with relative `_BASE_` inheritance,
* trailing `KEY VALUE` command-line overrides with dotted keys,
* `freeze()` / `defrost()` / `clone()` / `convert_to_dict()`.

The same YAML files the reference ships load unchanged.  One reference quirk is
handled on purpose: the Lyft overlays set `IMAGE.H` / `IMAGE.W`, which do not
exist in the defaults (fiery/configs/lyft/baseline.yml:15-16) - yacs would raise
on them; they are accepted here and stored, nothing reads them.
"""
import argparse
import ast
import copy
import os

import yaml

# keys the shipped reference YAMLs contain although the defaults do not
_TOLERATED_UNKNOWN_KEYS = {'IMAGE.H', 'IMAGE.W'}


class CfgNode(dict):
    """dict with attribute access, nesting and an immutability switch."""

    _FROZEN = '__frozen__'

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for key, value in (init or {}).items():
            dict.__setitem__(self, key, CfgNode(value) if isinstance(value, dict) else value)

    # -- attribute protocol -------------------------------------------------
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f'Attempted to set {name} to {value}, but the config is frozen')
        self[name] = CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value

    def __setitem__(self, key, value):
        if self.is_frozen():
            raise AttributeError(f'Attempted to set {key} to {value}, but the config is frozen')
        dict.__setitem__(self, key, value)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for key, value in self.items():
            dict.__setitem__(out, key, copy.deepcopy(value, memo))
        object.__setattr__(out, CfgNode._FROZEN, self.is_frozen())
        return out

    def __reduce__(self):
        return (CfgNode, (self.convert_to_dict(),))

    # -- immutability ---------------------------------------------------------
    def is_frozen(self):
        return self.__dict__.get(CfgNode._FROZEN, False)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for value in self.values():
            if isinstance(value, CfgNode):
                value._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    # -- merging ----------------------------------------------------------------
    def merge_from_other_cfg(self, other):
        _merge_into(self, other, [])

    def merge_from_file(self, path):
        _merge_into(self, _load_yaml_with_base(path), [])

    def merge_from_list(self, opts):
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError(f'Override list must be KEY VALUE pairs, got {opts}')
        for full_key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full_key.split('.')
            for part in parts[:-1]:
                if part not in node or not isinstance(node[part], CfgNode):
                    raise KeyError(f'Non-existent config key: {full_key}')
                node = node[part]
            leaf = parts[-1]
            if leaf not in node and full_key not in _TOLERATED_UNKNOWN_KEYS:
                raise KeyError(f'Non-existent config key: {full_key}')
            value = _decode(raw)
            if leaf in node:
                value = _coerce(value, node[leaf], full_key)
            node[leaf] = value

    def convert_to_dict(self):
        return convert_to_dict(self)

    def dump(self, **kwargs):
        return yaml.safe_dump(self.convert_to_dict(), **kwargs)


CN = CfgNode


def convert_to_dict(cfg_node, key_list=()):
    """Plain-dict copy of a config tree (reference: fiery/config.py:5-20)."""
    if not isinstance(cfg_node, CfgNode):
        return cfg_node
    return {k: convert_to_dict(v, tuple(key_list) + (k,)) for k, v in cfg_node.items()}


def _decode(raw):
    if not isinstance(raw, str):
        return raw
    try:
        return ast.literal_eval(raw)
    except (ValueError, SyntaxError):
        return raw


def _coerce(value, current, full_key):
    """yacs-style type check: the new value must be compatible with the default's type."""
    if current is None or value is None or isinstance(value, type(current)):
        return value
    if isinstance(current, tuple) and isinstance(value, list):
        return tuple(value)
    if isinstance(current, list) and isinstance(value, tuple):
        return list(value)
    if isinstance(current, float) and isinstance(value, int) and not isinstance(value, bool):
        return float(value)
    if isinstance(current, CfgNode) and isinstance(value, dict):
        return CfgNode(value)
    raise ValueError(
        f'Type mismatch ({type(current)} vs. {type(value)}) with values ({current} vs. {value}) '
        f'for config key: {full_key}')


def _merge_into(dst, src, trail):
    for key, value in src.items():
        full_key = '.'.join(trail + [key])
        if key not in dst:
            if full_key in _TOLERATED_UNKNOWN_KEYS:
                dict.__setitem__(dst, key, copy.deepcopy(value))
                continue
            raise KeyError(f'Non-existent config key: {full_key}')
        if isinstance(dst[key], CfgNode):
            if not isinstance(value, dict):
                raise ValueError(f'Config key {full_key} is a node, got {type(value)}')
            _merge_into(dst[key], value, trail + [key])
        else:
            if dst.is_frozen():
                raise AttributeError(f'Attempted to set {full_key}, but the config is frozen')
            dict.__setitem__(dst, key, _coerce(_decode(copy.deepcopy(value)), dst[key], full_key))


def _load_yaml_with_base(path):
    with open(path, 'r') as handle:
        data = yaml.safe_load(handle) or {}
    base = data.pop('_BASE_', None)
    if base is None:
        return data
    base = os.path.expanduser(base)
    if not os.path.isabs(base):
        base = os.path.join(os.path.dirname(path), base)
    merged = _load_yaml_with_base(base)
    _overlay(merged, data)
    return merged


def _overlay(dst, src):
    for key, value in src.items():
        if isinstance(value, dict) and isinstance(dst.get(key), dict):
            _overlay(dst[key], value)
        else:
            dst[key] = value


def _defaults():
    """Default tree; values restate fiery/config.py:32-123."""
    return CfgNode({
        'LOG_DIR': 'tensorboard_logs',
        'TAG': 'default',
        'GPUS': [0],
        'PRECISION': 32,
        'BATCHSIZE': 3,
        'EPOCHS': 20,
        'N_WORKERS': 5,
        'VIS_INTERVAL': 5000,
        'LOGGING_INTERVAL': 500,
        'PRETRAINED': {'LOAD_WEIGHTS': False, 'PATH': ''},
        'DATASET': {
            'DATAROOT': './nuscenes/', 'VERSION': 'trainval', 'NAME': 'nuscenes',
            'IGNORE_INDEX': 255, 'FILTER_INVISIBLE_VEHICLES': True,
        },
        'TIME_RECEPTIVE_FIELD': 3,
        'N_FUTURE_FRAMES': 4,
        'IMAGE': {
            'FINAL_DIM': (224, 480), 'RESIZE_SCALE': 0.3, 'TOP_CROP': 46,
            'ORIGINAL_HEIGHT': 900, 'ORIGINAL_WIDTH': 1600,
            'NAMES': ['CAM_FRONT_LEFT', 'CAM_FRONT', 'CAM_FRONT_RIGHT',
                      'CAM_BACK_LEFT', 'CAM_BACK', 'CAM_BACK_RIGHT'],
        },
        'LIFT': {
            'X_BOUND': [-50.0, 50.0, 0.5], 'Y_BOUND': [-50.0, 50.0, 0.5],
            'Z_BOUND': [-10.0, 10.0, 20.0], 'D_BOUND': [2.0, 50.0, 1.0],
        },
        'MODEL': {
            'ENCODER': {'DOWNSAMPLE': 8, 'NAME': 'efficientnet-b4', 'OUT_CHANNELS': 64,
                        'USE_DEPTH_DISTRIBUTION': True},
            'TEMPORAL_MODEL': {'NAME': 'temporal_block', 'START_OUT_CHANNELS': 64,
                               'EXTRA_IN_CHANNELS': 0, 'INBETWEEN_LAYERS': 0,
                               'PYRAMID_POOLING': True, 'INPUT_EGOPOSE': True},
            'DISTRIBUTION': {'LATENT_DIM': 32, 'MIN_LOG_SIGMA': -5.0, 'MAX_LOG_SIGMA': 5.0},
            'FUTURE_PRED': {'N_GRU_BLOCKS': 3, 'N_RES_LAYERS': 3},
            'DECODER': {},
            'BN_MOMENTUM': 0.1,
            'SUBSAMPLE': False,
        },
        'SEMANTIC_SEG': {'WEIGHTS': [1.0, 2.0], 'USE_TOP_K': True, 'TOP_K_RATIO': 0.25},
        'INSTANCE_SEG': {},
        'INSTANCE_FLOW': {'ENABLED': True},
        'PROBABILISTIC': {'ENABLED': True, 'WEIGHT': 100.0, 'FUTURE_DIM': 6},
        'FUTURE_DISCOUNT': 0.95,
        'OPTIMIZER': {'LR': 3e-4, 'WEIGHT_DECAY': 1e-7},
        'GRAD_NORM_CLIP': 5,
    })


_C = _defaults()

# Overlays equivalent to the reference's shipped YAML files, keyed by the path under
# fiery/configs/ (used when the YAML files themselves are not around, e.g. on the GPU box).
PRESETS = {
    'baseline.yml': {
        'TAG': 'baseline', 'GPUS': [0, 1, 2, 3], 'BATCHSIZE': 3, 'PRECISION': 16,
        'TIME_RECEPTIVE_FIELD': 3, 'N_FUTURE_FRAMES': 4, 'PROBABILISTIC': {'ENABLED': True},
        'MODEL': {'BN_MOMENTUM': 0.05, 'TEMPORAL_MODEL': {'NAME': 'temporal_block', 'INPUT_EGOPOSE': True}},
        'INSTANCE_FLOW': {'ENABLED': True}, 'OPTIMIZER': {'LR': 3e-4}, 'N_WORKERS': 10,
    },
    'single_timeframe.yml': {
        'TAG': 'single_timeframe_model', 'GPUS': [0, 1], 'BATCHSIZE': 8,
        'TIME_RECEPTIVE_FIELD': 1, 'N_FUTURE_FRAMES': 0, 'PROBABILISTIC': {'ENABLED': False},
        'MODEL': {'TEMPORAL_MODEL': {'NAME': 'identity', 'INPUT_EGOPOSE': False}},
        'INSTANCE_FLOW': {'ENABLED': False}, 'OPTIMIZER': {'LR': 1e-3}, 'N_WORKERS': 10,
    },
    'temporal_single_timeframe.yml': {
        '_BASE_': 'single_timeframe.yml', 'TAG': 'temporal_single_timeframe', 'BATCHSIZE': 4,
        'PRECISION': 16, 'TIME_RECEPTIVE_FIELD': 3,
        'MODEL': {'BN_MOMENTUM': 0.05, 'TEMPORAL_MODEL': {'NAME': 'temporal_block', 'INPUT_EGOPOSE': True}},
    },
    'literature/static_lss_setting.yml': {
        '_BASE_': 'single_timeframe.yml', 'TAG': 'lift_splat_setting',
        'DATASET': {'FILTER_INVISIBLE_VEHICLES': False},
    },
    'literature/static_pon_setting.yml': {
        '_BASE_': 'literature/static_lss_setting.yml', 'TAG': 'pyramid_occupancy_network_setting',
        'LIFT': {'X_BOUND': [-50.0, 50.0, 0.25], 'Y_BOUND': [-25.0, 25.0, 0.25]},
    },
    'literature/lift_splat_setting.yml': {
        '_BASE_': 'temporal_single_timeframe.yml', 'TAG': 'temporal_lift_splat_setting',
        'DATASET': {'FILTER_INVISIBLE_VEHICLES': False},
    },
    'literature/pon_setting.yml': {
        '_BASE_': 'temporal_single_timeframe.yml', 'TAG': 'temporal_pon_setting',
        'DATASET': {'FILTER_INVISIBLE_VEHICLES': False},
        'LIFT': {'X_BOUND': [-50.0, 50.0, 0.25], 'Y_BOUND': [-25.0, 25.0, 0.25]},
    },
    'literature/fishing_setting.yml': {
        '_BASE_': 'baseline.yml', 'TAG': 'fishing_setting', 'BATCHSIZE': 3,
        'DATASET': {'FILTER_INVISIBLE_VEHICLES': False},
        'LIFT': {'D_BOUND': [2.0, 16.0, 0.5], 'X_BOUND': [-16.0, 16.0, 0.1], 'Y_BOUND': [-9.6, 9.7, 0.1]},
    },
    'lyft/baseline.yml': {
        '_BASE_': 'baseline.yml', 'TAG': 'lyft_baseline', 'GPUS': [0, 1, 2, 3], 'BATCHSIZE': 3,
        'TIME_RECEPTIVE_FIELD': 5, 'N_FUTURE_FRAMES': 10, 'DATASET': {'NAME': 'lyft'},
        'IMAGE': {'H': 1080, 'W': 1920, 'RESIZE_SCALE': 0.25}, 'MODEL': {'SUBSAMPLE': True},
    },
    'lyft/single_timeframe.yml': {
        '_BASE_': 'single_timeframe.yml', 'TAG': 'lyft_single_frame', 'DATASET': {'NAME': 'lyft'},
        'IMAGE': {'H': 1080, 'W': 1920, 'RESIZE_SCALE': 0.25},
    },
    'debug_baseline.yml': {
        '_BASE_': 'baseline.yml', 'TAG': 'debug', 'EPOCHS': 2, 'BATCHSIZE': 1, 'GPUS': [0],
        'LOGGING_INTERVAL': 10, 'DATASET': {'VERSION': 'mini'}, 'VIS_INTERVAL': 8,
    },
    'lyft/debug_lyft.yml': {
        '_BASE_': 'lyft/baseline.yml', 'TAG': 'debug', 'BATCHSIZE': 1, 'GPUS': [0],
        'DATASET': {'VERSION': 'mini'}, 'VIS_INTERVAL': 4, 'N_WORKERS': 0,
    },
}


def _preset_overlay(name):
    if name not in PRESETS:
        raise KeyError(f'Unknown preset {name!r}; known: {sorted(PRESETS)}')
    data = copy.deepcopy(PRESETS[name])
    base = data.pop('_BASE_', None)
    if base is None:
        return data
    merged = _preset_overlay(base)
    _overlay(merged, data)
    return merged


def get_preset_cfg(name, opts=None, freeze=False):
    """Config for one of the reference's shipped YAML names without needing the file."""
    cfg = _C.clone()
    _merge_into(cfg, _preset_overlay(name), [])
    cfg.merge_from_list(opts)
    if freeze:
        cfg.freeze()
    return cfg


def get_parser():
    parser = argparse.ArgumentParser(description='Fiery training')
    parser.add_argument('--config-file', default='', metavar='FILE', help='path to config file')
    parser.add_argument('opts', help='Modify config options using the command-line',
                        default=None, nargs=argparse.REMAINDER)
    return parser


def get_cfg(args=None, cfg_dict=None):
    """Defaults, then `cfg_dict`, then `args.config_file` + `args.opts`; frozen when args are given."""
    cfg = _C.clone()
    if cfg_dict is not None:
        cfg.merge_from_other_cfg(CfgNode(cfg_dict))
    if args is not None:
        if args.config_file:
            cfg.merge_from_file(args.config_file)
        cfg.merge_from_list(args.opts)
        cfg.freeze()
    return cfg
