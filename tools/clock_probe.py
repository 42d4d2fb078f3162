#!/usr/bin/env python
"""Tuning aid: engine clock while conv launches of different sizes run back to back (rocm-smi polled from a thread)."""
import subprocess
import sys
import threading
import time
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiery_amd import native                                   # noqa: E402
from fiery_amd.ops import Buf, ConvOp, identity_chan_map       # noqa: E402

DEV = 'cuda:0'
lib = native.get()
samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            line = ' | '.join(l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l)
            samples.append(line)
        except Exception as e:                                  # noqa: BLE001
            samples.append(repr(e))
        time.sleep(0.3)


for n, H, W in ((1, 96, 226), (3, 200, 200), (6, 192, 256)):
    x = Buf(torch.randn(n, H, W, 128, device=DEV), n, H, W, 128)
    w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
    op = ConvOp(lib, w, identity_chan_map(128), (16, 0), torch.ones(128), torch.zeros(128), DEV, act=native.ACT_RELU)
    out = Buf.alloc(n, H, W, 128, DEV)
    op([x], out)
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    t = threading.Thread(target=poll)
    t.start()
    t0 = time.time()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    reps = 0
    while time.time() - t0 < 4.0:
        for _ in range(50):
            op([x], out)
        reps += 50
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    stop = True
    t.join()
    print(f'conv 128->128 n={n} {H}x{W}: {s.elapsed_time(e) * 1e3 / reps:.1f} us per launch over {reps} launches')
    for line in samples[2:6]:
        print('    ', line[:260])
