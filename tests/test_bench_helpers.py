"""bench.py's host-side helpers (no GPU): the counter-traffic summary the JSON line quotes, and the argument defaults the driver's
contract names (N = 1, K / W that finish within minutes)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_names_file_commit_and_prepass():
    bench = _bench()
    conv = bench.pmc_traffic(['k_conv_igemm (all tile shapes)'])
    pool = bench.pmc_traffic(['k_voxel_pool', 'fiery::k_rank_columns'])
    assert conv['file'].startswith('profiles/r') and conv['bytes_per_launch'] > 50e6
    # the pooling op's traffic counts the prepass too, and is what the op really moves: within 1.3x of 1,193 MB algorithmic
    assert set(pool['per_kernel']) == {'k_voxel_pool', 'fiery::k_rank_columns'}
    assert 1.0 < pool['bytes_per_launch'] / 1193.3e6 < 1.3
    table = json.load(open(os.path.join(ROOT, conv['file'])))
    assert conv['measured_at_commit'] == table.get('_measured_at_commit')
    assert conv['commit'], 'the summary must say which commit it belongs to (git log here, its own stamp on a GPU box)'
    assert bench.pmc_traffic(['no such kernel']) is None


def test_default_arguments_are_the_contract(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    args = bench.parse()
    assert args.gpus == 1 and args.steps == 10 and args.warmup == 3 and args.batch == 3 and args.config == 'baseline.yml'
    assert args.precision == 'f32' and args.layout == 'batch'


def test_parity_rows_compare_every_sample_and_print_both_bars():
    """The driver line's parity block: the max over ALL samples (round 4's compared sample 0 only), with the literal 1e-4 bar and
    the scaled bar side by side - an output can pass the second and miss the first, and the line must say so."""
    import torch
    bench = _bench()
    want = {'segmentation': torch.zeros(3, 2, 4, 4), 'flow': 8.0 * torch.ones(3, 2, 4, 4), 'absent': None}
    got = {'segmentation': want['segmentation'].clone(), 'flow': want['flow'].clone(), 'absent': None}
    got['segmentation'][2, 0, 1, 1] = 1.06e-4            # only the LAST sample misses the literal bar
    got['flow'][1, 1, 0, 0] += 3e-4                      # inside 1e-4 * |ref|_inf = 8e-4, outside the literal bar
    rows = bench.parity_rows(got, want)
    assert set(rows) == {'segmentation', 'flow'}
    seg, flow = rows['segmentation'], rows['flow']
    assert not seg['within_1e-4'] and not seg['within_scaled'] and seg['max_abs_err_per_sample'][:2] == [0.0, 0.0]
    assert seg['elements_within_1e-4'] == seg['elements'] - 1
    assert not flow['within_1e-4'] and flow['within_scaled'] and flow['ref_abs_max'] == 8.0
