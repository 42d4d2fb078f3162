import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native
from fiery_amd.ops import Buf, ConvOp, identity_chan_map
lib = native.get()
torch.manual_seed(0)
for (n, H, W) in ((1, 8, 8), (1, 16, 16), (2, 20, 20), (3, 200, 200)):
    cin = cout = 64
    x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
    w = torch.randn(cout, cin, 3, 3) / 24
    op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_RELU, tune=True)
    ref, out = Buf.alloc(n, H, W, cout, 'cuda'), Buf.alloc(n, H, W, cout, 'cuda')
    op.force_form = 128; op([x], ref)
    for env in ('0', '1'):
        os.environ['FIERY_WINOGRAD_GENERAL_EPILOGUE'] = env
        out.tensor.fill_(float('nan'))
        op.force_form = 'wino'; op([x], out)
        torch.cuda.synchronize()
        d = (out.tensor - ref.tensor).abs()
        bad = (d > 1e-4) | torch.isnan(d)
        print(n, H, W, 'general' if env == '1' else 'per-kind', 'max diff', d.nan_to_num(99).max().item(), 'bad elems', int(bad.sum()), 'of', d.numel())
        if bad.any():
            idx = bad.nonzero()[:6].tolist()
            print('   first bad (img,y,x,c):', idx, ' bad per channel-quad:', bad.view(-1, cout // 4, 4).any(-1).sum(0).tolist()[:16])
            print('   bad per (y%2, x%2):', [[int(bad[:, a::2, b::2].sum()) for b in (0, 1)] for a in (0, 1)])

print('---- where do the bad values come from? (1 x 8 x 8)')
n, H, W, cin, cout = 1, 8, 8, 64, 64
x = Buf(torch.randn(n, H, W, cin, device='cuda'), n, H, W, cin)
w = torch.randn(cout, cin, 3, 3) / 24
op = ConvOp(lib, w, identity_chan_map(cin), (cin // 8, 0), torch.ones(cout), torch.zeros(cout), 'cuda', act=native.ACT_NONE, tune=True)
ref, out = Buf.alloc(n, H, W, cout, 'cuda'), Buf.alloc(n, H, W, cout, 'cuda')
op.force_form = 128; op([x], ref)
os.environ['FIERY_WINOGRAD_GENERAL_EPILOGUE'] = '0'
op.force_form = 'wino'; op([x], out)
torch.cuda.synchronize()
r, o = ref.tensor[0].cpu(), out.tensor[0].cpu()
bad = ((o - r).abs() > 1e-4).nonzero().tolist()
for (y, xx, c) in bad[:8]:
    v = o[y, xx, c].item()
    near = ((r - v).abs() < 2e-5).nonzero().tolist()
    d = v - r[y, xx, c].item()
    # is the error a multiple of some other reference value?
    print(f'({y},{xx},{c}) got {v:+.5f} want {r[y, xx, c].item():+.5f} err {d:+.5f}; ref has this value at {near[:4]}; '
          f'err/ref(same px, c-1)={d / r[y, xx, c - 1].item():+.3f} err/ref(c+1)={d / r[y, xx, c + 1].item():+.3f}')
