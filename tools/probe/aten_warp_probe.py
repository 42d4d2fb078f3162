"""Which fp32 operations do ATen's CPU affine_grid / grid_sample fuse, and in which order?  Bit-compares numpy restatements
(with / without FMA, orders of summation) against the operators on THIS machine's CPU.  csrc/warp.hip follows the variants
that print True here (linspace halves with FMA, BLAS product k ascending, fused un-normalisation, fused four-term chain)."""

print('--- linspace / base grid'); 
import torch, numpy as np, math
torch.manual_seed(0)
f32=np.float32
def fma(a,b,c):
    return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)  # double-rounding risk negligible? products of f32 exact in f64; sum rounding to f64 then f32: rare double rounding
for W in (200,400,192,320,13):
    ls = torch.linspace(-1,1,W).numpy()
    i = np.arange(W)
    step = f32(2.0)/f32(W-1)
    a = (f32(-1)+step*i.astype(f32)).astype(f32)
    b = (f32(1)-step*(W-1-i).astype(f32)).astype(f32)
    half = W//2
    cand = np.where(i<half, a, b)
    print(W,'linspace sym nofma', np.array_equal(cand,ls))
    a2 = fma(np.full(W,step,f32), i.astype(f32), np.full(W,-1,f32))
    b2 = fma(np.full(W,-step,f32), (W-1-i).astype(f32), np.full(W,1,f32))
    cand2=np.where(i<half,a2,b2)
    print(W,'linspace sym fma', np.array_equal(cand2,ls), np.abs(cand2-ls).max())
    rng = (torch.linspace(-1,1,W)*(W-1)/W).numpy()
    c3 = ((ls*f32(W-1)).astype(f32)/f32(W)).astype(f32)
    print(W,'range mul-div', np.array_equal(c3,rng))

print('--- affine_grid product')
import torch, numpy as np, itertools
f32=np.float32
def fma(a,b,c):
    return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)
def mul(a,b): return (a*b).astype(f32)
def add(a,b): return (a+b).astype(f32)
torch.manual_seed(1)
for (H,W) in ((200,200),(400,200),(192,320)):
  for N in (1,2,3):
    th = torch.tensor([[0.9998,-0.0199,0.0123],[0.0199,0.9998,-0.0567]]).repeat(N,1,1)+0.001*torch.randn(N,2,3)
    grid = torch.nn.functional.affine_grid(th,(N,4,H,W),align_corners=False).numpy()
    xs = (torch.linspace(-1,1,W)*(W-1)/W).numpy(); ys=(torch.linspace(-1,1,H)*(H-1)/H).numpy()
    X = np.broadcast_to(xs[None,:],(H,W)); Y=np.broadcast_to(ys[:,None],(H,W))
    t = th.numpy()
    ok = {}
    for n in range(N):
      for o in range(2):
        a,b,c = (np.full((H,W),t[n,o,k],f32) for k in range(3))
        one = np.ones((H,W),f32)
        cands = {
          'nofma xyz': add(add(mul(X,a),mul(Y,b)),c),
          'fma k-asc': fma(one,c,fma(Y,b,mul(X,a))),
          'fma k-desc': fma(X,a,fma(Y,b,c)),
          'fma c first': fma(Y,b,fma(X,a,c)),
          'fma x last2': fma(X,a,add(mul(Y,b),c)),
        }
        for k,v in cands.items():
            ok.setdefault(k,True); ok[k] &= np.array_equal(v, grid[n,...,o])
    print(H,W,N,ok)

print('--- grid_sample')
import torch, numpy as np, itertools
f32=np.float32
def fma(a,b,c):
    return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)
def mul(a,b): return (a*b).astype(f32)
def add(a,b): return (a+b).astype(f32)
def sub(a,b): return (a-b).astype(f32)
torch.manual_seed(1)
for (H,W) in ((200,200),(400,200),(37,53)):
    N=2; C=3
    th = torch.tensor([[0.9998,-0.0199,0.0123],[0.0199,0.9998,-0.0567]]).repeat(N,1,1)+0.001*torch.randn(N,2,3)
    x = torch.randn(N,C,H,W)
    grid = torch.nn.functional.affine_grid(th,(N,C,H,W),align_corners=False)
    out = torch.nn.functional.grid_sample(x,grid,mode='bilinear',padding_mode='zeros',align_corners=False).numpy()
    g = grid.numpy(); xn = x.numpy()
    gx,gy = g[...,0],g[...,1]
    one=f32(1)
    res={}
    for un in ('nofma','fma'):
        if un=='nofma':
            fx = sub(mul(add(gx,one),f32(W/2)),f32(0.5)); fy = sub(mul(add(gy,one),f32(H/2)),f32(0.5))
        else:
            fx = fma(add(gx,one),np.full_like(gx,f32(W/2)),np.full_like(gx,f32(-0.5))); fy = fma(add(gy,one),np.full_like(gy,f32(H/2)),np.full_like(gy,f32(-0.5)))
        x0=np.floor(fx); y0=np.floor(fy)
        w=sub(fx,x0); e=sub(one,w); n=sub(fy,y0); s=sub(one,n)
        nw=mul(s,e); ne=mul(s,w); sw=mul(n,e); se=mul(n,w)
        ix=x0.astype(np.int64); iy=y0.astype(np.int64)
        def gat(iy,ix):
            m=(ix>=0)&(ix<W)&(iy>=0)&(iy<H)
            v=np.zeros((N,C,H,W),f32)
            for b in range(N):
                iyc=np.clip(iy[b],0,H-1); ixc=np.clip(ix[b],0,W-1)
                v[b]=np.where(m[b][None],xn[b][:,iyc,ixc],0)
            return v
        vnw=gat(iy,ix); vne=gat(iy,ix+1); vsw=gat(iy+1,ix); vse=gat(iy+1,ix+1)
        B=lambda a:np.broadcast_to(a[:,None],(N,C,H,W)).astype(f32)
        nw,ne,sw,se=B(nw),B(ne),B(sw),B(se)
        cands={
          'sum nofma': add(add(add(mul(vnw,nw),mul(vne,ne)),mul(vsw,sw)),mul(vse,se)),
          'sum fma chain': fma(vse,se,fma(vsw,sw,fma(vne,ne,mul(vnw,nw)))),
          'sum fma pair': fma(vse,se,fma(vsw,sw,add(mul(vnw,nw),mul(vne,ne)))),
          'sum fma b': fma(vse,se,add(fma(vne,ne,mul(vnw,nw)),mul(vsw,sw))),
        }
        for k,v in cands.items():
            res[(un,k)] = (np.array_equal(v,out), float(np.abs(v-out).max()))
    print(H,W); [print('  ',k,v) for k,v in res.items()]
