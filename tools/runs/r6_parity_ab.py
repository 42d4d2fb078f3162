#!/usr/bin/env python
"""Round 6: where does the distance to the reference come from?  baseline.yml at batch 3 (BASELINE.json configs[1]), three input
draws, every output against the oracle's - under switches that change OUR rounding: the pooling sums in 64-bit fixed point
(FIERY_POOL_DETERMINISTIC), the GRU gates' sigmoid through libm's expf and the correctly rounded division (a library built with
-DFIERY_SIGMOID_LIBM=1), the direct form instead of Winograd (FIERY_CONV_WINOGRAD=0).
  python tools/runs/r6_parity_ab.py oracle <dir>            the oracle's outputs (and the float64 evaluation's) -> <dir>/want_*.pt
  python tools/runs/r6_parity_ab.py gpu <dir> [fixed]       this process's library / switches against them
(The oracle is test infrastructure: this script is a measurement tool, not a product path.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.model import Fiery                              # noqa: E402
from fiery_amd.synthetic import make_inputs, make_lifted_features, randomise_weights   # noqa: E402

DRAWS = ((0, 100), (0, 1), (7, 8))
mode, out_dir = sys.argv[1], sys.argv[2]
cfg = get_preset_cfg('baseline.yml')
torch.manual_seed(0)
model = Fiery(cfg).eval()
sd = randomise_weights(model)
B, n = 3, 6
rf, nf, D = model.receptive_field, model.n_future, model.depth_channels


def inputs(s_in, s_lift):
    _, K, E, ego = make_inputs(B, rf + nf, n, with_image=False, seed=s_in)
    _, _, lifted = make_lifted_features(B * rf * n, 64, D, (28, 60), seed=s_lift)
    return lifted.view(B, rf, n, 64, D, 28, 60), K, E, ego


if mode == 'oracle':
    from oracle import bev_stack
    os.makedirs(out_dir, exist_ok=True)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    for s_in, s_lift in DRAWS:
        lifted, K, E, ego = inputs(s_in, s_lift)
        with torch.no_grad():
            want = bev_stack.bev_hot_path(sd_cpu, cfg, lifted, K, E, ego)
            exact = bev_stack.bev_hot_path_exact(sd_cpu, cfg, lifted, K, E, ego) if os.environ.get('EXACT', '1') == '1' else None
        torch.save({'want': want, 'exact': exact}, os.path.join(out_dir, f'want_{s_in}_{s_lift}.pt'))
        if exact is not None:
            print(f'draw {s_in}/{s_lift}: reference vs float64  ' + '  '.join(
                f'{k} {(want[k].double() - exact[k].double()).abs().max().item():.3e}' for k in want if want[k] is not None), flush=True)
else:
    from fiery_amd import native
    dev = torch.device('cuda:0')
    model = model.to(dev)
    model.camera_matrix_mode = 'device'
    if len(sys.argv) > 3 and sys.argv[3] == 'fixed':
        model.engine().pool_flags = native.POOL_DETERMINISTIC
        for e in getattr(model, '_sample_engines', {}).values():
            e.pool_flags = native.POOL_DETERMINISTIC
    worst = {}
    for s_in, s_lift in DRAWS:
        ref = torch.load(os.path.join(out_dir, f'want_{s_in}_{s_lift}.pt'))
        lifted, K, E, ego = inputs(s_in, s_lift)
        with torch.no_grad():
            got = model.bev_forward(lifted.to(dev), K.to(dev), E.to(dev), ego.to(dev))
        row = []
        for k, v in ref['want'].items():
            if v is None:
                continue
            err = (got[k].float().cpu() - v).abs().max().item()
            ex = (got[k].double().cpu() - ref['exact'][k].double()).abs().max().item() if ref['exact'] is not None else float('nan')
            worst[k] = max(worst.get(k, 0.0), err)
            row.append(f'{k} {err:.3e} (vs f64 {ex:.2e})')
        print(f'  draw {s_in}/{s_lift}: ' + '  '.join(row), flush=True)
    print('  WORST over draws: ' + '  '.join(f'{k} {v:.3e}{"" if v <= 1e-4 else " !"}' for k, v in worst.items()), flush=True)
