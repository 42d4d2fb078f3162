"""Diagnostic: voxel pooling (plain and under autograd) on a 104 x 104 grid on the GPU, step by step."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiery_amd.config import get_preset_cfg
from fiery_amd.model import Fiery
from fiery_amd.train_graph import TrainGraph
from fiery_amd import native
from fiery_amd.synthetic import make_inputs, make_lifted_features
lib = native.get()
cfg = get_preset_cfg('baseline.yml', ['LIFT.X_BOUND', '[-26.0, 26.0, 0.5]', 'LIFT.Y_BOUND', '[-26.0, 26.0, 0.5]', 'N_FUTURE_FRAMES', '2'])
m = Fiery(cfg).cuda()
_, K, E, ego = [None if t is None else t.cuda() for t in make_inputs(1, 5, 6, with_image=False)]
_, _, lifted = make_lifted_features(18, 64, m.depth_channels, (28, 60), seed=1)
lifted = lifted.view(1, 3, 6, 64, m.depth_channels, 28, 60).cuda()
tg = TrainGraph(m, lib)
print('start', flush=True)
with torch.no_grad():
    bev = tg._pooled(K[:, :3].contiguous(), E[:, :3].contiguous(), lifted)
torch.cuda.synchronize(); print('plain ok', bev.abs().sum().item(), flush=True)
leaf = lifted.clone().requires_grad_()
bev = tg._pooled(K[:, :3].contiguous(), E[:, :3].contiguous(), leaf)
torch.cuda.synchronize(); print('autograd fwd ok', bev.abs().sum().item(), flush=True)
bev.sum().backward()
torch.cuda.synchronize(); print('bwd ok', leaf.grad.abs().sum().item(), flush=True)
b64 = bev.detach().double()
torch.cuda.synchronize(); print('double ok', flush=True)
del bev, leaf, b64
import gc; gc.collect(); torch.cuda.empty_cache(); print('freed ok', flush=True)
