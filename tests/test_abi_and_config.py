"""Host-side checks: the C-ABI library exports every symbol include/fiery_hip.h declares (load only, no
compute), and the configuration surface accepts what the reference's does."""
import os
import re

import pytest

from fiery_amd import native
from fiery_amd.config import CfgNode, get_cfg, get_parser, get_preset_cfg, PRESETS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'fiery_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fiery_[a-z0-9_]+)\s*\(', text)))


def test_header_and_bindings_agree():
    assert _declared_symbols() == sorted(native.EXPORTED_SYMBOLS)


def test_library_builds_and_exports_every_declared_symbol():
    from fiery_amd.build import build
    path = build(verbose=False)                    # hipcc cross-compiles gfx950 without a GPU
    import ctypes
    dll = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(dll, name), name
    assert dll.fiery_abi_version() == native.ABI_VERSION
    native.Lib(path)                               # full signature binding


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(native.NativeError, match='no fallback'):
        native.Lib(str(tmp_path / 'libfiery_hip.so'))


def test_presets_and_overrides():
    cfg = get_preset_cfg('literature/pon_setting.yml', ['BATCHSIZE', '5', 'LIFT.X_BOUND', '[-10.0, 10.0, 0.5]'])
    assert cfg.BATCHSIZE == 5 and cfg.LIFT.X_BOUND == [-10.0, 10.0, 0.5] and cfg.LIFT.Y_BOUND == [-25.0, 25.0, 0.25]
    assert cfg.MODEL.TEMPORAL_MODEL.NAME == 'temporal_block' and cfg.N_FUTURE_FRAMES == 0
    with pytest.raises(KeyError):
        get_preset_cfg('baseline.yml', ['NOT.A.KEY', '1'])
    with pytest.raises(ValueError):
        get_preset_cfg('baseline.yml', ['BATCHSIZE', 'abc'])


def test_yaml_with_base_and_freeze(tmp_path):
    (tmp_path / 'base.yml').write_text("TAG: 'b'\nOPTIMIZER:\n  LR: 3e-4\nMODEL:\n  BN_MOMENTUM: 0.05\n")
    (tmp_path / 'sub').mkdir()
    (tmp_path / 'sub' / 'child.yml').write_text("_BASE_: '../base.yml'\nTAG: 'c'\nIMAGE:\n  H: 1080\n")
    args = get_parser().parse_args(['--config-file', str(tmp_path / 'sub' / 'child.yml'), 'GPUS', '[0, 1]'])
    cfg = get_cfg(args)
    assert cfg.TAG == 'c' and cfg.OPTIMIZER.LR == 3e-4 and cfg.MODEL.BN_MOMENTUM == 0.05 and cfg.GPUS == [0, 1]
    assert cfg.IMAGE.H == 1080                      # the Lyft YAML quirk is tolerated
    with pytest.raises(AttributeError):
        cfg.BATCHSIZE = 1
    d = cfg.convert_to_dict()
    assert isinstance(d, dict) and not isinstance(d['MODEL'], CfgNode)
    again = get_cfg(cfg_dict=d)                      # trainer.py:21 rebuilds the cfg from the plain dict
    assert again.convert_to_dict() == d
    again.GPUS = '[0]'                               # evaluate.py:28 mutates the unfrozen cfg in place


@pytest.mark.needs_reference
def test_presets_equal_the_reference_yaml_files():
    base = '/root/reference/fiery/configs'
    for name in PRESETS:
        args = get_parser().parse_args(['--config-file', os.path.join(base, name)])
        assert get_cfg(args).convert_to_dict() == get_preset_cfg(name).convert_to_dict(), name


def test_fork_guard_of_the_input_pipeline_entry_points(monkeypatch):
    """`require_usable_gpu_process` (labels / image preparation inside a forked DataLoader worker): raises the message with
    the remedies when torch reports a bad fork - and only then."""
    import torch
    from fiery_amd import native
    monkeypatch.setattr(torch.cuda, '_is_in_bad_fork', lambda: False, raising=False)
    native.require_usable_gpu_process('instance labels')
    monkeypatch.setattr(torch.cuda, '_is_in_bad_fork', lambda: True, raising=False)
    with pytest.raises(RuntimeError, match="multiprocessing_context='spawn'"):
        native.require_usable_gpu_process('instance labels')


def test_padded_rows_mark_survives_detach_in_the_training_graph():
    """`_pixel_major` widens a channel slice of padded rows in place instead of copying; the mark that says so is a Python
    attribute, which `detach()` drops - `_detached` carries it over."""
    import torch
    from fiery_amd import train_graph as tg
    rows = torch.zeros(2, 5, 6, 40)
    x = tg._padded_rows(rows[..., :35].permute(0, 3, 1, 2), 35, 40)
    assert getattr(x, '_fiery_padded_rows', 0) == 40
    assert getattr(x.detach(), '_fiery_padded_rows', 0) == 0
    t = tg._pixel_major(tg._detached(x))
    assert t.shape == (2, 5, 6, 40) and t.data_ptr() == rows.data_ptr()            # widened in place: no copy
