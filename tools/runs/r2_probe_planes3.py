"""Tuning aid (round 2): the row-dealt slice pattern of the pooling unit with more loads in flight, non-temporal loads and
rolling re-requests (tools/probe/hbm_probe.hip k_read_planes3)."""
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
probe.probe_planes3.argtypes = [C.c_void_p, C.c_longlong] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
x = torch.randn(9 * 6 * 64 * 48 * 28 * 60 // 4, 4, device='cuda')
sink = torch.zeros(4, device='cuda')
nbytes = x.numel() * 4
stream = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = []
for lds, threads in ((160000, 1024), (80000, 512), (80000, 384), (54000, 256), (54000, 384)):
    for slices, nt, rolling in ((1, 0, 0), (1, 1, 0), (2, 0, 0), (2, 1, 0), (4, 0, 0), (1, 0, 1), (1, 1, 1)):
        cases.append((lds, threads, slices, nt, rolling))
for lds, threads, slices, nt, rolling in cases:
    for _ in range(2):
        probe.probe_planes3(x.data_ptr(), nbytes, 576, threads, slices, nt, rolling, lds, sink.data_ptr(), stream)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        probe.probe_planes3(x.data_ptr(), nbytes, 576, threads, slices, nt, rolling, lds, sink.data_ptr(), stream)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f'planes3: LDS {lds:6d} B ({163840 // lds}/CU), {threads:4d} thr, slices in flight {slices}, nt={nt}, rolling={rolling}: '
          f'{us:7.1f} us -> {nbytes / us / 1e3:7.1f} GB/s', flush=True)
