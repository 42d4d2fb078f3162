"""Diagnostic: graph_step of tests/test_train_graph.py stage by stage on the GPU."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from fiery_amd.train_graph import TrainGraph
from fiery_amd import native
from tests.helpers import forward_case, randomise_weights
lib = native.get()
mode = sys.argv[1]
B = 2 if mode.endswith('B2') else 1
if 'nocudnn' in mode:
    torch.backends.cudnn.enabled = False
cfg = get_preset_cfg('baseline.yml', ['LIFT.X_BOUND', '[-26.0, 26.0, 0.5]', 'LIFT.Y_BOUND', '[-26.0, 26.0, 0.5]', 'N_FUTURE_FRAMES', '2'])
torch.manual_seed(0)
model = Fiery(cfg)
state = {k: v.clone() for k, v in randomise_weights(model).items()}
lifted, K, E, ego, labels, noise = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, B, 6, with_labels=True, with_noise=True)
dtype = torch.float64 if mode in ('exact', 'nowarp', 'stackonly') else torch.float32
conv = None if mode.startswith('hip') else (lambda x, w, s, p, lib: F.conv2d(x, w, None, s, p))
def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)
model = Fiery(cfg); model.load_state_dict(state); model = model.train().to(device='cuda', dtype=dtype); model._lib = lib
pooler = Fiery(cfg).cuda(); pooler._lib = lib
rf = model.receptive_field
leaf = lifted.clone().cuda().requires_grad_()
bev = TrainGraph(pooler, lib)._pooled(K[:, :rf].contiguous().cuda(), E[:, :rf].contiguous().cuda(), leaf[:, :rf])
say('pooled')
cast = lambda t: None if t is None else t.to(device='cuda', dtype=dtype)
graph = TrainGraph(model, lib, conv2d=conv)
if mode == 'nowarp':
    import fiery_amd.train_graph as TG
    TG.cumulative_warp_features = lambda x, *a: x
x = cast(bev)
say('cast')
out = graph.bev_stack(x, cast(ego[:, :rf]), cast(labels), cast(noise))
say('forward')
g = torch.Generator().manual_seed(123)
total = 0.0
for k in sorted(out):
    if out[k] is not None:
        total = total + (out[k] * torch.randn(out[k].shape, generator=g).to(device='cuda', dtype=dtype)).sum()
say('loss', total.item())
if mode == 'stackonly':
    grads = torch.autograd.grad(total, [p for n, p in model.named_parameters() if not n.startswith('encoder.')], allow_unused=True)
    say('param grads only ok')
else:
    total.backward()
    say('backward')
    print(leaf.grad.abs().sum().item())
