"""Tuning aid: kernel breakdown of the image trunk on the engine (54 images, 224x480)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from tests.helpers import randomise_weights
from torch.profiler import profile, ProfilerActivity
cfg = get_preset_cfg('baseline.yml'); torch.manual_seed(0)
model = Fiery(cfg).eval(); randomise_weights(model); model = model.cuda()
x = torch.randn(54, 3, 224, 480, device='cuda')
eng = model.engine()
with torch.no_grad():
    for _ in range(3): eng.trunk_endpoints(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        eng.trunk_endpoints(x); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
