"""Tuning aid (round 2): the (channel, frame) pooling unit's bare read pattern when its LDS footprint lets 1, 2 or 3
workgroups share a CU (tools/probe/hbm_probe.hip k_read_planes2)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, 'tools', 'probe', 'libhbm_probe.so')
subprocess.check_call(['hipcc', '-w', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       os.path.join(ROOT, 'tools', 'probe', 'hbm_probe.hip'), '-o', so])
probe = C.CDLL(so)
probe.probe_planes2.argtypes = [C.c_void_p, C.c_longlong] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
x = torch.randn(9 * 6 * 64 * 48 * 28 * 60 // 4, 4, device='cuda')
sink = torch.zeros(4, device='cuda')
nbytes = x.numel() * 4
stream = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = [(160000, 1024, 7, 1), (160000, 1024, 7, 4), (80000, 512, 7, 1), (80000, 768, 7, 1), (80000, 1024, 7, 1), (80000, 512, 7, 4),
         (80000, 768, 7, 4), (54000, 256, 7, 1), (54000, 384, 7, 1), (54000, 512, 7, 1), (54000, 512, 14, 1), (54000, 384, 14, 1),
         (54000, 256, 28, 1), (54000, 256, 7, 4), (54000, 384, 7, 4), (54000, 512, 7, 4), (40000, 256, 7, 1), (40000, 256, 7, 4),
         (40000, 384, 7, 4)]
for lds, threads, batch, R in cases:
    for _ in range(2):
        probe.probe_planes2(x.data_ptr(), nbytes, 576, threads, batch, R, lds, sink.data_ptr(), stream)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        probe.probe_planes2(x.data_ptr(), nbytes, 576, threads, batch, R, lds, sink.data_ptr(), stream)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f'planes2: 576 units, LDS {lds:6d} B ({163840 // lds} per CU), {threads:4d} threads, batch {batch:2d}, R={R}: '
          f'{us:7.1f} us -> {nbytes / us / 1e3:7.1f} GB/s', flush=True)
