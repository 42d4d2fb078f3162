"""Tuning aid: per-launch time and effective bandwidth of the image trunk's convolutions on the engine."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fiery_amd import ops
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from tests.helpers import randomise_weights
cfg = get_preset_cfg('baseline.yml'); torch.manual_seed(0)
model = Fiery(cfg).eval(); randomise_weights(model); model = model.cuda()
x = torch.randn(54, 3, 224, 480, device='cuda')
eng = model.engine()
with torch.no_grad():
    for _ in range(3): eng.trunk_endpoints(x)
    torch.cuda.synchronize()
    ops.PROFILE_SINK = []
    eng.trunk_endpoints(x)
    torch.cuda.synchronize()
recs, ops.PROFILE_SINK = ops.PROFILE_SINK, None
tot = 0
for kind, s, e, w, d in recs:
    us = s.elapsed_time(e) * 1e3
    tot += us
    kT, kH, kW, stride, cin, cout, n, H, W = d
    mb = 4.0 * n * H * W * (cin * stride * stride + cout) / 1e6
    print(f'{kH}x{kW} s{stride} {cin:4d}->{cout:4d} {n}x{H}x{W}: {us:7.1f} us  {w / us / 1e6:6.1f} TFLOP/s  {mb / us * 1e3 / 1e3:7.1f} GB/s (in+out {mb:.0f} MB)')
print('total conv us', tot)
