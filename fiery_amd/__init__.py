"""fiery_amd: MI355X-native (gfx950) implementation of the FIERY camera-to-BEV hot path.

`Fiery` mirrors `fiery.models.fiery.Fiery` of wayveai/fiery; its BEV path runs on the hand-written HIP
kernels in `fiery_amd/csrc` (libfiery_hip.so, C ABI in include/fiery_hip.h).
"""
from .config import get_cfg, get_parser, get_preset_cfg, CfgNode   # noqa: F401


def __getattr__(name):
    if name == 'Fiery':
        from .model import Fiery
        return Fiery
    raise AttributeError(name)
