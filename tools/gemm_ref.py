#!/usr/bin/env python
"""Tuning aid: what the vendor library (rocBLAS / hipBLASLt through torch.matmul, fp32) reaches on this box for
GEMMs of the convolution shapes - the practical ceiling of v_mfma_f32_32x32x2_f32 kernels here."""
import torch

DEV = 'cuda:0'


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


torch.backends.cuda.matmul.allow_tf32 = False
for m, k, n in ((120000, 1152, 128), (147456, 1152, 128), (120000, 1152, 64), (600000, 576, 256), (8192, 8192, 8192),
                (16384, 4096, 4096), (120000, 288, 32)):
    a = torch.randn(m, k, device=DEV)
    b = torch.randn(k, n, device=DEV)
    us = timed(lambda: a @ b)
    print(f'torch.matmul fp32 {m}x{k} @ {k}x{n}: {us:9.1f} us  {2.0 * m * k * n / us / 1e6:6.1f} TFLOP/s', flush=True)
