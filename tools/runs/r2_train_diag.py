"""Diagnostic: the fp64 evaluation of the training graph on the host and on the GPU (which of them works on the box)."""
import sys, os, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from fiery_amd.train_graph import TrainGraph
from fiery_amd.synthetic import randomise_weights
dev = sys.argv[1]
cfg = get_preset_cfg('baseline.yml', ['LIFT.X_BOUND', '[-26.0, 26.0, 0.5]', 'LIFT.Y_BOUND', '[-26.0, 26.0, 0.5]', 'N_FUTURE_FRAMES', '2'])
torch.manual_seed(0)
m = Fiery(cfg); randomise_weights(m); m.train()
dt = torch.float64
m = m.to(device=dev, dtype=dt)
tg = TrainGraph(m, lib=object(), conv2d=lambda x, w, s, p, lib: F.conv2d(x, w, None, s, p))
bev = torch.randn(1, 3, 64, 104, 104, dtype=dt, device=dev).requires_grad_()
ego = torch.zeros(1, 3, 6, dtype=dt, device=dev); ego[..., 0] = 1.0
labels = torch.randn(1, 3, 6, 104, 104, dtype=dt, device=dev)
t0 = time.time()
out = tg.bev_stack(bev, ego, labels, torch.randn(1, 1, 32, dtype=dt, device=dev))
print(dev, 'fwd ok', time.time() - t0, flush=True)
sum(v.sum() for v in out.values() if v is not None).backward()
if dev == 'cuda': torch.cuda.synchronize()
print(dev, 'bwd ok', time.time() - t0, torch.get_num_threads(), flush=True)
