"""TEST INFRASTRUCTURE: a second, independent restatement of the two third-party networks the reference binds.

The reference takes its image trunk from `efficientnet-pytorch==0.7.0` (fiery/models/encoder.py:2,16,40-91;
environment.yml:16) and its decoder backbone from `torchvision==0.8.1` `resnet18` (fiery/models/decoder.py:2,10-17;
environment.yml:9).  Neither package is installed offline, so the product carries its own restatement of both
(`fiery_amd/backbone.py`, `fiery_amd/modules.py:BasicBlockWeights`).  If the checker used those same classes, product and
checker would share one definition and a mistake in it would cancel out.  This module is therefore written separately,
from the published architectures, and imports nothing from `fiery_amd`:

* `oracle/ref_shims.py` installs THESE classes as `efficientnet_pytorch.EfficientNet` / `torchvision.models.resnet.resnet18`
  under the reference's own, unmodified code (the reference side of every fixture and of the `needs_reference` tests);
* `tests/test_third_party_restatements.py` compares them with the product's restatements (same `state_dict` keys and
  shapes, same outputs for the same weights).

PARITY UNPINNED against the real packages (absent); what this buys is that two independently written statements of the
published networks agree.

EfficientNet (Tan & Le, 2019) as efficientnet-pytorch builds it: stem 3x3/2 -> MBConv stages -> (head, unused here);
b0 stage table scaled by width 1.4 / depth 1.8 for b4 (channels rounded to multiples of 8, never below 90 % of the scaled
value; repeats rounded up); BatchNorm eps 1e-3, momentum 0.01; swish; squeeze-excite ratio 0.25 of the block's INPUT
channels; identity skip (with drop-connect in training) when stride 1 and in == out; `from_pretrained` uses convolutions
with TensorFlow "SAME" padding evaluated once for the nominal input size of the model (380 for b4) as it shrinks stage by
stage - so the padding is a property of the layer, not of the tensor that comes in.
resnet18 stages (He et al., 2015) as torchvision builds them: BasicBlock = conv3x3-BN-ReLU-conv3x3-BN, plus the identity
(or a 1x1/stride conv + BN of it when the shape changes), then ReLU; layer1 64 (stride 1), layer2 128 (stride 2), layer3
256 (stride 2), two blocks each; kaiming-normal (fan_out) convolutions, BN weight 1 / bias 0, and `zero_init_residual`
zeroes the last BN weight of every block.
"""
import math
from types import SimpleNamespace

import torch
from torch import nn
from torch.nn import functional as F


# ---------------------------------------------------------------------------------------------------------------------
# EfficientNet trunk
# ---------------------------------------------------------------------------------------------------------------------
class SamePaddedConv(nn.Module):
    """Conv2d preceded by the zero padding TensorFlow's SAME rule gives for a fixed nominal input size.

    Holds `weight` / `bias` directly (so its parameters are named `<name>.weight`, `<name>.bias` like a Conv2d's)."""

    def __init__(self, in_channels, out_channels, kernel_size, nominal_size, stride=1, groups=1, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.groups = (kernel_size, kernel_size), (stride, stride), groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.weight[0].numel())
            nn.init.uniform_(self.bias, -bound, bound)
        # SAME: the output has ceil(size / stride) positions; whatever the window overhangs is split, the odd cell after
        out_size = -(-nominal_size // stride)
        total = max((out_size - 1) * stride + kernel_size - nominal_size, 0)
        before = total // 2
        self.padding = (before, total - before, before, total - before)              # left, right, top, bottom
        self.static_padding = SimpleNamespace(padding=self.padding)                 # (the attribute callers inspect)

    def forward(self, x):
        if any(self.padding):
            x = F.pad(x, self.padding)
        return F.conv2d(x, self.weight, self.bias, stride=self.stride, groups=self.groups)


def scaled_channels(channels, width, divisor=8):
    scaled = channels * width
    rounded = max(divisor, int(scaled + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < 0.9 * scaled else rounded


class MBConv(nn.Module):
    """Mobile inverted bottleneck with squeeze-and-excitation; attribute names of efficientnet-pytorch's MBConvBlock."""

    def __init__(self, kernel, stride, expand_ratio, in_channels, out_channels, nominal_size):
        super().__init__()
        hidden = in_channels * expand_ratio
        self.has_expansion = expand_ratio != 1
        self.identity_skip = stride == 1 and in_channels == out_channels
        if self.has_expansion:
            self._expand_conv = SamePaddedConv(in_channels, hidden, 1, nominal_size)
            self._bn0 = nn.BatchNorm2d(hidden, eps=1e-3, momentum=0.01)
        self._depthwise_conv = SamePaddedConv(hidden, hidden, kernel, nominal_size, stride=stride, groups=hidden)
        self._bn1 = nn.BatchNorm2d(hidden, eps=1e-3, momentum=0.01)
        squeeze = max(1, int(in_channels * 0.25))
        self._se_reduce = SamePaddedConv(hidden, squeeze, 1, 1, bias=True)
        self._se_expand = SamePaddedConv(squeeze, hidden, 1, 1, bias=True)
        self._project_conv = SamePaddedConv(hidden, out_channels, 1, nominal_size)
        self._bn2 = nn.BatchNorm2d(out_channels, eps=1e-3, momentum=0.01)

    def forward(self, inputs, drop_connect_rate=None):
        y = inputs
        if self.has_expansion:
            y = self._bn0(self._expand_conv(y))
            y = y * torch.sigmoid(y)
        y = self._bn1(self._depthwise_conv(y))
        y = y * torch.sigmoid(y)
        squeezed = y.mean(dim=(2, 3), keepdim=True)
        squeezed = self._se_reduce(squeezed)
        squeezed = self._se_expand(squeezed * torch.sigmoid(squeezed))
        y = y * torch.sigmoid(squeezed)
        y = self._bn2(self._project_conv(y))
        if self.identity_skip:
            if self.training and drop_connect_rate:
                survive = 1.0 - drop_connect_rate
                keep = torch.floor(survive + torch.rand(y.shape[0], 1, 1, 1, dtype=y.dtype, device=y.device))
                y = y / survive * keep
            y = y + inputs
        return y


class EfficientNet(nn.Module):
    """The attribute surface fiery/models/encoder.py:40-91 uses: `_conv_stem`, `_bn0`, `_swish`, `_blocks`,
    `_global_params.drop_connect_rate`, and the deletable head `_conv_head, _bn1, _avg_pooling, _dropout, _fc`."""

    STAGES_B0 = [  # kernel, stride, expansion, in, out, repeats
        (3, 1, 1, 32, 16, 1), (3, 2, 6, 16, 24, 2), (5, 2, 6, 24, 40, 2), (3, 2, 6, 40, 80, 3), (5, 1, 6, 80, 112, 3),
        (5, 2, 6, 112, 192, 4), (3, 1, 6, 192, 320, 1)]
    COMPOUND = {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b4': (1.4, 1.8, 380, 0.4)}     # width, depth, size, dropout

    def __init__(self, name):
        super().__init__()
        width, depth, size, dropout = self.COMPOUND[name]
        self._global_params = SimpleNamespace(drop_connect_rate=0.2, image_size=size, dropout_rate=dropout)
        stem_channels = scaled_channels(32, width)
        self._conv_stem = SamePaddedConv(3, stem_channels, 3, size, stride=2)
        self._bn0 = nn.BatchNorm2d(stem_channels, eps=1e-3, momentum=0.01)
        self._swish = _Swish()
        size = -(-size // 2)
        blocks = []
        for kernel, stride, expansion, c_in, c_out, repeats in self.STAGES_B0:
            c_in, c_out = scaled_channels(c_in, width), scaled_channels(c_out, width)
            blocks.append(MBConv(kernel, stride, expansion, c_in, c_out, size))
            size = -(-size // stride)
            for _ in range(int(math.ceil(depth * repeats)) - 1):
                blocks.append(MBConv(kernel, 1, expansion, c_out, c_out, size))
        self._blocks = nn.ModuleList(blocks)
        head_channels = scaled_channels(1280, width)
        self._conv_head = SamePaddedConv(blocks[-1]._project_conv.out_channels, head_channels, 1, size)
        self._bn1 = nn.BatchNorm2d(head_channels, eps=1e-3, momentum=0.01)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(dropout)
        self._fc = nn.Linear(head_channels, 1000)

    @classmethod
    def from_pretrained(cls, name, **_):
        return cls(name)             # ImageNet weights need a download; real weights come with the FIERY checkpoint

    @classmethod
    def from_name(cls, name, **_):
        return cls(name)


class _Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


# ---------------------------------------------------------------------------------------------------------------------
# resnet18 stages
# ---------------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.downsample = None
        if stride != 1 or in_channels != out_channels:
            self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False),
                                            nn.BatchNorm2d(out_channels))

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + shortcut)


def resnet18(pretrained=False, zero_init_residual=False):
    """An object with the members fiery/models/decoder.py:10-17 takes: bn1, relu, layer1, layer2, layer3."""
    assert not pretrained
    net = SimpleNamespace(bn1=nn.BatchNorm2d(64), relu=nn.ReLU(inplace=True))
    width_in = 64
    for index, (width, stride) in enumerate(((64, 1), (128, 2), (256, 2)), start=1):
        setattr(net, f'layer{index}', nn.Sequential(BasicBlock(width_in, width, stride), BasicBlock(width, width, 1)))
        width_in = width
    for part in (net.bn1, net.layer1, net.layer2, net.layer3):
        for module in part.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(module, nn.BatchNorm2d):
                nn.init.ones_(module.weight)
                nn.init.zeros_(module.bias)
    if zero_init_residual:
        for layer in (net.layer1, net.layer2, net.layer3):
            for block in layer:
                nn.init.zeros_(block.bn2.weight)
    return net


# ---------------------------------------------------------------------------------------------------------------------
# The reference's use of the trunk, on THIS file's network (checker side; nothing here imports fiery_amd)
# ---------------------------------------------------------------------------------------------------------------------
def encoder_endpoints(net, x, downsample=8, version='b4'):
    """fiery/models/encoder.py:58-86: stem, blocks with their drop-connect rates, the feature maps kept at every change
    of resolution -> (deep, shallow) = the two levels the lift head fuses."""
    endpoints = []
    x = net._swish(net._bn0(net._conv_stem(x)))
    previous = x
    n_blocks = len(net._blocks)
    for index, block in enumerate(net._blocks):
        rate = net._global_params.drop_connect_rate
        if rate:
            rate *= float(index) / n_blocks
        x = block(x, drop_connect_rate=rate)
        if previous.size(2) > x.size(2):
            endpoints.append(previous)
        previous = x
        if downsample == 8 and ((version == 'b0' and index == 10) or (version == 'b4' and index == 21)):
            break
    endpoints.append(x)
    return (endpoints[4], endpoints[3]) if downsample == 16 else (endpoints[3], endpoints[2])


def lift_head(sd, deep, shallow, prefix='encoder.'):
    """fiery/models/encoder.py:87-96 + fiery/layers/convolutions.py:182-200 from a state_dict: x2 bilinear of the deep
    level, concatenation behind the shallow one, two 3x3 conv + BatchNorm + ReLU, the 1x1 depth layer (with bias)."""
    import torch.nn.functional as F
    x = torch.cat([shallow, F.interpolate(deep, scale_factor=2, mode='bilinear', align_corners=False)], dim=1)
    for conv, bn in ((0, 1), (3, 4)):
        p = f'{prefix}upsampling_layer.conv.'
        x = F.conv2d(x, sd[f'{p}{conv}.weight'], None, 1, 1)
        x = F.batch_norm(x, sd[f'{p}{bn}.running_mean'], sd[f'{p}{bn}.running_var'], sd[f'{p}{bn}.weight'], sd[f'{p}{bn}.bias'],
                         False, 0.0, 1e-5)
        x = F.relu(x)
    return F.conv2d(x, sd[f'{prefix}depth_layer.weight'], sd[f'{prefix}depth_layer.bias'])
