#!/usr/bin/env python
"""GATED EXPERIMENT (round 5; CPU only, test-side: it runs the oracle): would Winograd F(2x2, 3x3) for the
3x3 / stride-1 convolutions of the BEV stack keep the fp32 parity configuration's accuracy?

F(2x2, 3x3) does 2.25x fewer multiplies than the direct form (16 per 2x2 outputs and channel pair instead of 36), with
add-only input / output transforms - on a part whose fp32 MFMAs run at vector-ALU rate that is the one algorithmic lever left
for the layers that hold ~85 % of the convolution time.  Before any kernel: the oracle's hot path (baseline.yml, configs[1])
with every 3x3 / stride 1 / pad 1 `conv2d` replaced by an fp32 Winograd emulation (transformed weights computed in fp64 and
rounded once, as a host-side packer would; input transform, channel contraction and output transform in fp32), against
  * the oracle itself (fp32 direct: what the parity tests compare with; the bar is 1e-4 max-abs), and
  * the fp64 evaluation of the same network (`bev_hot_path_exact`: the value both approximate).
Decision rule of the verdict: proceed only if the Winograd path's distance to fp64 stays <= the reference's own.

  python tests/experiments/winograd_numerics.py [batch]      -> prints a table, writes profiles/r5_winograd_numerics.json
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
BT = torch.tensor([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0]], dtype=torch.float64)
AT = torch.tensor([[1.0, 1.0, 1.0, 0.0], [0.0, 1.0, -1.0, -1.0]], dtype=torch.float64)
_real_conv2d = F.conv2d
COUNT = {'winograd': 0, 'direct': 0}


def winograd_conv2d(x, w, bias=None):
    """3x3, stride 1, pad 1, fp32: Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A per 2x2 output block."""
    n, c, h, wd = x.shape
    k = w.shape[0]
    U = (G @ w.double() @ G.t()).float()                                    # (k, c, 4, 4): host-side, rounded once
    hp, wp = (h + 1) // 2 * 2, (wd + 1) // 2 * 2                            # pad to whole 2x2 blocks
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    th, tw = hp // 2, wp // 2
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                  # (n, c, th, tw, 4, 4)
    bt = BT.float()
    V = torch.einsum('ij,nctujk,lk->nctuil', bt, d, bt)                      # B^T d B, adds only (entries 0 / +-1)
    M = torch.einsum('kcil,nctuil->nktuil', U, V)                            # channel contraction per transform point, fp32
    at = AT.float()
    Y = torch.einsum('ij,nktujl,ml->nktuim', at, M, at)                      # (n, k, th, tw, 2, 2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, k, hp, wp)[:, :, :h, :wd]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y.contiguous()


def patched_conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    s = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    if x.dtype == torch.float32 and w.shape[2:] == (3, 3) and s == 1 and pd == 1 and groups == 1 and dilation == 1:
        COUNT['winograd'] += 1
        return winograd_conv2d(x, w, bias)
    COUNT['direct'] += 1
    return _real_conv2d(x, w, bias, stride, padding, dilation, groups)


def main():
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    from fiery_amd.synthetic import randomise_weights
    from oracle import bev_stack
    from tests.helpers import forward_case
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg).eval()
    sd = {k: v.cpu() for k, v in randomise_weights(model).items()}
    lifted, K, E, ego, _, _ = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, B, 6)
    # a quick self-check of the emulation on one layer
    xt, wt = torch.randn(1, 16, 9, 11), torch.randn(8, 16, 3, 3) / 12
    assert (winograd_conv2d(xt, wt) - _real_conv2d(xt, wt, padding=1)).abs().max() < 1e-5
    with torch.no_grad():
        direct = bev_stack.bev_hot_path(sd, cfg, lifted, K, E, ego)
        exact = bev_stack.bev_hot_path_exact(sd, cfg, lifted, K, E, ego)
        F.conv2d = patched_conv2d
        try:
            wino = bev_stack.bev_hot_path(sd, cfg, lifted, K, E, ego)
        finally:
            F.conv2d = _real_conv2d
    rows = {}
    print(f'baseline.yml batch {B}: {COUNT["winograd"]} conv2d calls through Winograd F(2x2, 3x3), {COUNT["direct"]} direct')
    print(f'{"output":22s} {"|wino - direct|":>16s} {"|direct - fp64|":>16s} {"|wino - fp64|":>16s} {"ref |.|max":>10s}   1e-4 bar vs direct')
    for k, v in direct.items():
        if v is None:
            continue
        e = exact[k].float() if exact[k].dtype != torch.float32 else exact[k]
        a, b, c = (wino[k] - v).abs().max().item(), (v.double() - exact[k]).abs().max().item(), (wino[k].double() - exact[k]).abs().max().item()
        rows[k] = dict(wino_vs_direct=a, direct_vs_fp64=b, wino_vs_fp64=c, ref_abs_max=v.abs().max().item())
        print(f'{k:22s} {a:16.3e} {b:16.3e} {c:16.3e} {v.abs().max().item():10.3f}   {"inside" if a <= 1e-4 else "OUTSIDE"}')
    worst_ratio = max(r['wino_vs_fp64'] / max(r['direct_vs_fp64'], 1e-12) for r in rows.values())
    verdict = ('Winograd stays as close to fp64 as the direct form' if worst_ratio <= 1.0 else
               f'Winograd is up to {worst_ratio:.2f}x further from fp64 than the direct fp32 form')
    print(verdict)
    out = dict(config='baseline.yml', batch=B, conv2d_calls=COUNT, outputs=rows, worst_ratio_to_direct_distance=worst_ratio, verdict=verdict,
               what='oracle hot path with every 3x3/s1/p1 conv2d through an fp32 Winograd F(2x2,3x3) emulation; max-abs differences')
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r5_winograd_numerics.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
