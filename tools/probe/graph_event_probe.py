"""Probe: do timing events recorded INSIDE a captured hipGraph give elapsed times on replay?  (bench.py could then bracket the
pooling op in the served mode.)"""
import torch
dev = 'cuda:0'
a = torch.randn(64 << 20, device=dev)
s = torch.cuda.Stream(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
b = a * 2
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        b = a * 2
        e0.record()
        c = b + a
        e1.record()
        d = c + 1
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('elapsed in graph', e0.elapsed_time(e1) * 1e3, 'us (c = b + a over 64 M floats: ~150 us expected)')
except Exception as ex:                                            # noqa: BLE001
    print('ERR', repr(ex)[:300])
