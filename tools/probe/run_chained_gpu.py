"""GPU run of tools/probe/chained_gemm_swapped.hip: the register-chained second product on real MFMA lanes (layout check vs fp64)."""
import ctypes as C
import os
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
dll = C.CDLL(os.path.join(HERE, '_bin', 'libchained.so'))


def cout_of(r, hi):
    return 8 * (r // 4) + 4 * hi + r % 4


rng = np.random.default_rng(7)
for n_pixels, K1 in ((96, 64), (120000, 288)):
    A1 = rng.standard_normal((n_pixels, K1)).astype(np.float32)
    W1 = (rng.standard_normal((K1, 32)) / np.sqrt(K1)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    shift = (rng.standard_normal(32) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((32, 64)) / np.sqrt(32)).astype(np.float32)
    W2p = np.stack([W2[cout_of(r, hi)] for r in range(16) for hi in range(2)]).astype(np.float32)
    dev = [torch.from_numpy(a).cuda() for a in (A1, W1, scale, shift, W2p)]
    O = torch.full((n_pixels, 64), float('nan'), device='cuda')
    p = lambda t: C.c_void_p(t.data_ptr())
    torch.cuda.synchronize()
    rc = dll.probe_chained_swapped(*[p(t) for t in dev], p(O), n_pixels, K1)
    torch.cuda.synchronize()
    h = np.maximum((A1.astype(np.float64) @ W1.astype(np.float64)) * scale + shift, 0.0)
    want = h @ W2.astype(np.float64)
    got = O.cpu().numpy()
    print(f'n_pixels {n_pixels} K1 {K1}: rc {rc}, finite {np.isfinite(got).all()}, max |err| {np.abs(got - want).max():.3e} '
          f'(scale {np.abs(want).max():.2f})')
