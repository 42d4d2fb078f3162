"""Tuning aid (round 5): `Fiery.forward` from images as bench.py times it - hipGraph replays - for rocprofv3 --kernel-trace."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from fiery_amd.synthetic import make_inputs
from tests.helpers import randomise_weights
cfg = get_preset_cfg('baseline.yml'); torch.manual_seed(0)
model = Fiery(cfg).eval(); randomise_weights(model); model = model.cuda()
B, rf, n = 3, model.receptive_field, 6
image, K, E, ego = [t.cuda() for t in make_inputs(B, rf + model.n_future, n, image_hw=tuple(cfg.IMAGE.FINAL_DIM), seed=0)]
with torch.no_grad():
    for _ in range(3):
        model.forward_graph(image, K, E, ego)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.forward_graph(image, K, E, ego)
    torch.cuda.synchronize()
print(f'from images, graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per step (B = {B}, {B * rf * n} images)')
