"""Image trunk used by the lift head.

The reference obtains its trunk from the third-party package `efficientnet-pytorch==0.7.0`
(reference: fiery/models/encoder.py:2,16; environment.yml:16), which is not installed offline and is
NOT part of the BEV hot path ("trunk out of scope").  This module restates
that package's published EfficientNet architecture (Tan & Le 2019; MBConv + squeeze-excite + swish,
"static same" padding computed for the nominal 380 px resolution of b4) with the package's
attribute and `state_dict` names (`_conv_stem`, `_bn0`, `_blocks.N._expand_conv`, ...), so that a FIERY
checkpoint's `encoder.backbone.*` keys load.  It runs on stock PyTorch-ROCm ops.

PARITY UNPINNED: the real package cannot be imported here and the reference has no test that touches
it, so equality with `efficientnet_pytorch` is by construction from the published architecture only.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

# (kernel, stride, expand, in, out, repeats) of EfficientNet-b0; scaled by width/depth below.
_B0_STAGES = ((3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2), (3, 2, 6, 40, 80, 3),
              (5, 1, 6, 80, 112, 3), (5, 2, 6, 112, 192, 4), (3, 1, 6, 192, 320, 1))
# name -> (width, depth, resolution, dropout)
_SCALING = {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b4': (1.4, 1.8, 380, 0.4)}


def _round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


class _Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class _SamePadConv(nn.Conv2d):
    """Convolution with TensorFlow-style 'same' padding fixed at construction for a given image size."""

    def __init__(self, cin, cout, kernel_size, image_size, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, kernel_size, stride=stride, groups=groups, bias=bias)
        ih = iw = image_size
        k, s = kernel_size, stride
        oh, ow = math.ceil(ih / s), math.ceil(iw / s)
        pad_h = max((oh - 1) * s + (k - 1) + 1 - ih, 0)
        pad_w = max((ow - 1) * s + (k - 1) + 1 - iw, 0)
        if pad_h > 0 or pad_w > 0:
            self.static_padding = nn.ZeroPad2d((pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2))
        else:
            self.static_padding = nn.Identity()

    def forward(self, x):
        return F.conv2d(self.static_padding(x), self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


def same_pads(conv):
    """(left, right, top, bottom) of a 'static same padding' convolution, read the way efficientnet-pytorch 0.7 publishes it
    (`conv.static_padding`: an `nn.ZeroPad2d` with a `.padding` 4-tuple, or an `nn.Identity`): works for the package's
    modules and for any restatement that keeps the attribute."""
    pads = getattr(getattr(conv, 'static_padding', None), 'padding', None)
    return tuple(int(v) for v in pads) if pads is not None else (0, 0, 0, 0)


def mbconv_geometry(blk):
    """(stride, input channels, output channels) of an MBConv block, read from its LAYERS - the names the checkpoint fixes
    (`_expand_conv`, `_depthwise_conv`, `_project_conv`) - not from bookkeeping attributes, which differ between the
    efficientnet-pytorch package (`_block_args`) and restatements of it.  The block has the identity skip iff
    stride == 1 and cin == cout (`id_skip` is set for every block of the published b0 / b4 schedules)."""
    dw = blk._depthwise_conv
    stride = dw.stride[0] if isinstance(dw.stride, (tuple, list)) else dw.stride
    first = blk._expand_conv if hasattr(blk, '_expand_conv') else dw
    return int(stride), int(first.in_channels), int(blk._project_conv.out_channels)


class MBConvBlock(nn.Module):
    def __init__(self, kernel, stride, expand, cin, cout, image_size, se_ratio=0.25):
        super().__init__()
        self.stride, self.cin, self.cout, self.expand = stride, cin, cout, expand
        mid = cin * expand
        if expand != 1:
            self._expand_conv = _SamePadConv(cin, mid, 1, image_size)
            self._bn0 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        self._depthwise_conv = _SamePadConv(mid, mid, kernel, image_size, stride=stride, groups=mid)
        self._bn1 = nn.BatchNorm2d(mid, momentum=0.01, eps=1e-3)
        squeezed = max(1, int(cin * se_ratio))
        self._se_reduce = _SamePadConv(mid, squeezed, 1, 1, bias=True)
        self._se_expand = _SamePadConv(squeezed, mid, 1, 1, bias=True)
        self._project_conv = _SamePadConv(mid, cout, 1, image_size)
        self._bn2 = nn.BatchNorm2d(cout, momentum=0.01, eps=1e-3)
        self._swish = _Swish()

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand != 1:
            x = self._swish(self._bn0(self._expand_conv(x)))
        x = self._swish(self._bn1(self._depthwise_conv(x)))
        gate = F.adaptive_avg_pool2d(x, 1)
        gate = self._se_expand(self._swish(self._se_reduce(gate)))
        x = torch.sigmoid(gate) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            if drop_connect_rate and self.training:
                keep = 1.0 - drop_connect_rate
                mask = torch.floor(keep + torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device))
                x = x / keep * mask
            x = x + inputs
        return x


class EfficientNet(nn.Module):
    """Trunk with the attribute surface fiery/models/encoder.py:40-91 relies on."""

    def __init__(self, name='efficientnet-b4'):
        super().__init__()
        if name not in _SCALING:
            raise ValueError(f'unsupported trunk {name}; known {sorted(_SCALING)}')
        width, depth, resolution, dropout = _SCALING[name]
        self._global_params = SimpleNamespace(drop_connect_rate=0.2, image_size=resolution, dropout_rate=dropout)
        stem = _round_filters(32, width)
        self._conv_stem = _SamePadConv(3, stem, 3, resolution, stride=2)
        self._bn0 = nn.BatchNorm2d(stem, momentum=0.01, eps=1e-3)
        size = math.ceil(resolution / 2)
        blocks = []
        for kernel, stride, expand, cin, cout, repeats in _B0_STAGES:
            cin, cout = _round_filters(cin, width), _round_filters(cout, width)
            for r in range(int(math.ceil(depth * repeats))):
                blocks.append(MBConvBlock(kernel, stride if r == 0 else 1, expand, cin if r == 0 else cout, cout, size))
                if r == 0:
                    size = math.ceil(size / stride)
        self._blocks = nn.ModuleList(blocks)
        head = _round_filters(1280, width)
        # present at construction like the package; the lift head deletes them (encoder.py:52-56)
        self._conv_head = _SamePadConv(blocks[-1].cout, head, 1, size)
        self._bn1 = nn.BatchNorm2d(head, momentum=0.01, eps=1e-3)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(dropout)
        self._fc = nn.Linear(head, 1000)
        self._swish = _Swish()

    @classmethod
    def from_pretrained(cls, name, **_):
        """Pretrained ImageNet weights need a download; offline the trunk is randomly initialised and
        real weights arrive through the FIERY checkpoint's `encoder.backbone.*` keys."""
        return cls(name)

    @classmethod
    def from_name(cls, name, **_):
        return cls(name)
