import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from tests.helpers import randomise_weights
def timed(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
cfg = get_preset_cfg('baseline.yml'); torch.manual_seed(0)
model = Fiery(cfg).eval(); randomise_weights(model); model = model.cuda()
x = torch.randn(54, 3, 224, 480, device='cuda')
enc = model.encoder
with torch.no_grad():
    print('default', timed(lambda: enc.trunk_endpoints(x)))
    torch.backends.cudnn.benchmark = True
    print('benchmark=True', timed(lambda: enc.trunk_endpoints(x)))
    enc.backbone.to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
    print('channels_last + benchmark', timed(lambda: enc.trunk_endpoints(xc)))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        enc.trunk_endpoints(xc); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))
