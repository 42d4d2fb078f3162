#!/usr/bin/env python
"""Round 6: per-shape table of the convolution launches of ONE eager one-stream pass of `Fiery.forward` from images (trunk + lift head +
hot path), with the other entry points by name - which layers of the trunk are slow, and in which form they ran."""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiery_amd import native, ops                              # noqa: E402
from fiery_amd.config import get_preset_cfg                    # noqa: E402
from fiery_amd.model import Fiery                              # noqa: E402
from fiery_amd.synthetic import make_inputs, randomise_weights  # noqa: E402

dev = torch.device('cuda:0')
cfg = get_preset_cfg('baseline.yml')
torch.manual_seed(0)
model = Fiery(cfg).eval()
randomise_weights(model)
model = model.to(dev)
model.camera_matrix_mode = 'device'
model.sample_streams = False
B, n = 3, 6
image, K, E, ego = make_inputs(B, model.receptive_field + model.n_future, n, image_hw=tuple(cfg.IMAGE.FINAL_DIM), seed=0)
args = [t.to(dev) for t in (image, K, E, ego)]
shapes = []
lib = native.get()
for meth, pick in (('depthwise_conv', lambda a: ('depthwise', a[5], a[3], a[4], f'k{a[8]} s{a[9]}')),
                   ('se_gate_nhwc', lambda a: ('se_gate', a[5], a[4], 0, '')),
                   ('scale_channels', lambda a: ('scale', a[4], a[3], 0, ''))):
    orig = getattr(lib, meth)

    def wrapped(*a, _orig=orig, _pick=pick):
        if native.CALL_SINK is None:
            return _orig(*a)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        r_ = _orig(*a)
        e_.record()
        shapes.append((_pick(a), s_, e_))
        return r_
    setattr(lib, meth, wrapped)
with torch.no_grad():
    for _ in range(2):
        model(*args)
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)
    native.CALL_SINK, ops.PROFILE_SINK = [], []
    model(*args)
    torch.cuda.synchronize()
calls, convs = native.CALL_SINK, ops.PROFILE_SINK
native.CALL_SINK = ops.PROFILE_SINK = None
rows = defaultdict(lambda: [0.0, 0.0, 0])
for k, s, e, w, d in convs:
    if k == 'conv_igemm':
        r = rows[tuple(d)]
        r[0] += s.elapsed_time(e) * 1e3
        r[1] += w
        r[2] += 1
total = sum(r[0] for r in rows.values())
print(f'{"n":>3s} {"us":>9s} {"share":>6s} {"TF":>7s}  [kT, k, k, stride, cin, cout, images, Ho, Wo, form]   bytes-bound us @5TB/s')
for d, (us, w, cnt) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:45]:
    kT, kH, kW, st, cin, cout, nimg, Ho, Wo, form = d
    px = nimg * Ho * Wo
    nbytes = 4.0 * (px * st * st * cin + px * cout)
    print(f'{cnt:3d} {us:9.1f} {us / total:6.1%} {w / us / 1e6:7.1f}  {list(d)}   {nbytes / 5e6 * cnt:8.1f}')
print(f'convolutions {total / 1e3:.2f} ms')
other = defaultdict(lambda: [0.0, 0])
for name, s, e in calls:
    if name != 'fiery_conv_fwd':
        other[name][0] += s.elapsed_time(e) * 1e3
        other[name][1] += 1
for name, (us, cnt) in sorted(other.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f'{cnt:3d} {us:9.1f}  {name}')

print('depthwise / squeeze-excite launches (channels, rows or pixels, cols, kernel): us, bytes-bound us at 5 TB/s')
for (kind, c, h, w, ks), s_, e_ in shapes:
    us = s_.elapsed_time(e_) * 1e3
    if kind == 'depthwise':
        st = int(ks.split('s')[1])
        nb = 4.0 * 54 * c * (h * w + (h // st) * (w // st))
    elif kind == 'se_gate':
        nb = 4.0 * 54 * c * h
    else:
        nb = 8.0 * 54 * c * h
    print(f'  {kind:10s} C={c:4d} {h:6d} x {w:4d} {ks:6s} {us:8.1f} us   {nb / 5e6:7.1f}')
