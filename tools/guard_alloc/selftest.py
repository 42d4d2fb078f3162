"""TEST TOOL: does the guard allocator itself behave?  Device fills / copies to the host / copies from the host for tensors of
many sizes placed against the guard page (a copy engine that rounds its transfers could drop or fault on the tail)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

so = os.path.join(HERE, 'libguard_alloc.so')
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(so, 'guard_malloc', 'guard_free'))
bad = 0
for n in (1, 3, 63, 252 // 4, 1023, 4096 // 4, 4097, 241920, 1693440, 1451520, 2 * 64 * 200 * 200, 2 * 200 * 200 * 64 + 5):
    a = torch.arange(n, dtype=torch.float32, device='cuda') + 1.0
    dev_sum = a.double().sum().item()
    host = a.cpu()
    ok_d2h = bool((host == torch.arange(n, dtype=torch.float32) + 1.0).all())
    b = (torch.arange(n, dtype=torch.float32) * 2 + 1).cuda()
    ok_h2d = abs(b.double().sum().item() - float((torch.arange(n, dtype=torch.float64) * 2 + 1).sum())) < 1e-6 * max(1, n) ** 2
    last = b[-1:].cpu().item() == 2 * (n - 1) + 1
    c = a.clone()
    ok_d2d = bool((c == a).all().item())
    print(f'n={n:9d} ptr%4096={a.data_ptr() % 4096:5d}  d2h {ok_d2h}  h2d {ok_h2d} (last {last})  d2d {ok_d2d}  sum {dev_sum == n * (n + 1) / 2}')
    bad += not (ok_d2h and ok_h2d and last and ok_d2d)
print('selftest', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
