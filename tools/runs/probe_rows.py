"""Tuning aid: the pooling kernel's read pattern with a column's rows dealt out to R lane groups (tools/probe/hbm_probe.hip)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, 'tools', 'probe', 'libhbm_probe.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       os.path.join(ROOT, 'tools', 'probe', 'hbm_probe.hip'), '-o', so])
probe = C.CDLL(so)
probe.probe_read.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
x = torch.randn(9 * 6 * 64 * 48 * 28 * 60 // 4, 4, device='cuda')
sink = torch.zeros(4, device='cuda')
nbytes = x.numel() * 4
stream = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for blocks, threads in ((1152, 512), (576, 1024), (2304, 256)):
    for R in (1, 2, 4):
        for _ in range(2):
            probe.probe_read(x.data_ptr(), nbytes, blocks, threads, R, 4, sink.data_ptr(), stream)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            probe.probe_read(x.data_ptr(), nbytes, blocks, threads, R, 4, sink.data_ptr(), stream)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / reps
        print(f'rows dealt to R={R} lane groups, blocks={blocks} threads={threads}: {us:7.1f} us -> {nbytes / us / 1e3:7.1f} GB/s', flush=True)

for mode, name in ((5, 'non-temporal'), (6, 'plain')):
    for _ in range(2):
        probe.probe_read(x.data_ptr(), nbytes, 576, 1024, 7, mode, sink.data_ptr(), stream)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        probe.probe_read(x.data_ptr(), nbytes, 576, 1024, 7, mode, sink.data_ptr(), stream)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f'rows dealt: plane-kernel pattern (576 x 1024 threads, 160 KB LDS, 7 rows in flight, {name} loads): {us:7.1f} us -> {nbytes / us / 1e3:7.1f} GB/s', flush=True)
