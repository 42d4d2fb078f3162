import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native
from fiery_amd.ops import Buf, ConvOp, HeadsOut, identity_chan_map
lib = native.get()
for (n, hw, cin) in ((2, (9, 11), 16), (1, (16, 16), 64), (3, (200, 200), 64)):
    g = torch.Generator().manual_seed(5)
    n_outs, sig = [2, 1, 2, 2], [False, True, False, False]
    x = torch.randn(n, cin, *hw, generator=g)
    w3 = torch.randn(4 * 64, cin, 3, 3, generator=g) * 0.15 / (cin / 16) ** 0.5
    scale = torch.rand(4 * 64, generator=g) + 0.5
    shift = torch.randn(4 * 64, generator=g) * 0.3
    w1 = [torch.randn(k, 64, generator=g) * 0.2 for k in n_outs]
    b1 = [torch.randn(k, generator=g) for k in n_outs]
    xb = Buf(x.permute(0, 2, 3, 1).contiguous().cuda(), n, hw[0], hw[1], cin)
    groups = [h for h, k in enumerate(n_outs) for _ in range(k)]
    hidden = F.relu(F.conv2d(x, w3, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    for form in (0, 'wino'):
        for gen in ('0', '1'):
            if form == 0 and gen == '1':
                continue
            os.environ['FIERY_WINOGRAD_GENERAL_EPILOGUE'] = gen
            op = ConvOp(lib, w3, identity_chan_map(cin), (cin // 8, 0), scale, shift, 'cuda', act=native.ACT_RELU, tune=True)
            op.attach_heads(torch.cat(w1), torch.cat(b1), groups, [sig[h] for h in groups])
            op.force_form = form
            outs = [torch.full((n, k, *hw), float('nan'), device='cuda') for k in n_outs]
            hwp = hw[0] * hw[1]
            planes = [(outs[h].data_ptr() + 4 * j * hwp, n_outs[h] * hwp) for h in range(4) for j in range(n_outs[h])]
            op([xb], HeadsOut(n, *hw, outs[0]), head_planes=planes)
            torch.cuda.synchronize()
            errs = []
            for h in range(4):
                want = F.conv2d(hidden[:, 64 * h:64 * h + 64], w1[h].view(-1, 64, 1, 1), b1[h])
                if sig[h]:
                    want = torch.sigmoid(want)
                errs.append((outs[h].cpu() - want).abs().nan_to_num(99).max().item())
            print(n, hw, cin, 'form', form, 'general' if gen == '1' else 'per-kind', 'max err per head', ['%.2e' % e for e in errs])
