"""Parameter containers of the BEV stack.

These classes own the weights only: their attribute paths reproduce the reference's
`state_dict` key names and shapes one-for-one, so a released FIERY checkpoint loads with
`strict=True` (checkpoint contract).  None of them computes
anything: the arithmetic lives in the HIP kernels driven by `fiery_amd.engine`.

Key-name sources in the reference (file:line):
  fiery/layers/convolutions.py:9-61 (ConvBlock), :64-168 (Bottleneck), :203-214 (UpsamplingAdd),
  :182-200 (UpsamplingConcat); fiery/layers/temporal.py:10-25 (SpatialGRU), :65-85 (CausalConv3d),
  :107-117 (1x1x1 block), :167-215 (pyramid pooling), :218-265 (TemporalBlock), :120-164 (Bottleneck3D);
  fiery/models/temporal_model.py:6-45; fiery/models/future_prediction.py:7-25;
  fiery/models/distributions.py:7-56; fiery/models/decoder.py:7-51 (+ torchvision 0.8.1 resnet18
  BasicBlock naming: conv1/bn1/conv2/bn2/downsample.{0,1}).
"""
from collections import OrderedDict

import torch.nn as nn


def _named(**parts):
    return nn.Sequential(OrderedDict(parts))


def _bn_relu2d(channels):
    return nn.Sequential(nn.BatchNorm2d(channels), nn.ReLU(inplace=True))


class ConvNormAct2d(nn.Module):
    """`conv` [+ `norm`] [+ `activation`] holder; keys `conv.*`, `norm.*`."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, norm=True, relu=True, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding=(kernel_size - 1) // 2, bias=bias)
        self.norm = nn.BatchNorm2d(cout) if norm else None
        self.activation = nn.ReLU(inplace=True) if relu else None


class ResidualBottleneck(nn.Module):
    """1x1 squeeze -> 3x3 (optionally stride 2) -> 1x1 expand, with an identity or pooled skip."""

    def __init__(self, cin, cout=None, downsample=False):
        super().__init__()
        cout = cout or cin
        mid = int(cin / 2)
        self.downsample = downsample
        self.in_channels, self.mid_channels, self.out_channels = cin, mid, cout
        self.layers = _named(
            conv_down_project=nn.Conv2d(cin, mid, 1, bias=False),
            abn_down_project=_bn_relu2d(mid),
            conv=nn.Conv2d(mid, mid, 3, stride=2 if downsample else 1, padding=1, bias=False),
            abn=_bn_relu2d(mid),
            conv_up_project=nn.Conv2d(mid, cout, 1, bias=False),
            abn_up_project=_bn_relu2d(cout),
            dropout=nn.Dropout2d(p=0.0),
        )
        if cout == cin and not downsample:
            self.projection = None
        else:
            skip = OrderedDict()
            if downsample:
                skip['upsample_skip_proj'] = nn.MaxPool2d(kernel_size=2, stride=2)
            skip['conv_skip_proj'] = nn.Conv2d(cin, cout, 1, bias=False)
            skip['bn_skip_proj'] = nn.BatchNorm2d(cout)
            self.projection = nn.Sequential(skip)


class UpsampleAddWeights(nn.Module):
    """x2 bilinear -> 1x1 conv -> BN, added to a skip tensor; keys `upsample_layer.{1,2}.*`."""

    def __init__(self, cin, cout, scale_factor=2):
        super().__init__()
        self.upsample_layer = nn.Sequential(
            nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=False),
            nn.Conv2d(cin, cout, 1, bias=False),
            nn.BatchNorm2d(cout),
        )


class UpsampleConcatWeights(nn.Module):
    """x2 bilinear of the coarse map, concat, two 3x3 conv+BN+ReLU; keys `conv.{0,1,3,4}.*`."""

    def __init__(self, cin, cout, scale_factor=2):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=False)
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
        )


# ----------------------------------------------------------------------------------------------
# temporal model
# ----------------------------------------------------------------------------------------------
def _pointwise3d(cin, cout):
    return _named(conv=nn.Conv3d(cin, cout, 1, bias=False), norm=nn.BatchNorm3d(cout),
                  activation=nn.ReLU(inplace=True))


class CausalConv3dWeights(nn.Module):
    def __init__(self, cin, cout, kernel_size=(2, 3, 3)):
        super().__init__()
        kt, kh, kw = kernel_size
        self.kernel_size = tuple(kernel_size)
        self.pad = nn.ConstantPad3d(((kw - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kh - 1) // 2, kt - 1, 0), 0)
        self.conv = nn.Conv3d(cin, cout, kernel_size, bias=False)
        self.norm = nn.BatchNorm3d(cout)
        self.activation = nn.ReLU(inplace=True)


class PyramidPoolWeights(nn.Module):
    def __init__(self, cin, reduction, pool_sizes):
        super().__init__()
        self.pool_sizes = [tuple(p) for p in pool_sizes]
        branches = []
        for size in self.pool_sizes:
            assert size[0] == 2, 'time kernel of the pyramid pooling must be 2'
            branches.append(_named(
                avgpool=nn.AvgPool3d(kernel_size=size, stride=(1, *size[1:]), padding=(size[0] - 1, 0, 0),
                                     count_include_pad=False),
                conv_bn_relu=_pointwise3d(cin, reduction)))
        self.features = nn.ModuleList(branches)


class TemporalBlockWeights(nn.Module):
    def __init__(self, cin, cout=None, use_pyramid_pooling=False, pool_sizes=None):
        super().__init__()
        self.in_channels = cin
        self.half_channels = cin // 2
        self.out_channels = cout or cin
        self.kernels = [(2, 3, 3), (1, 3, 3)]
        self.use_pyramid_pooling = use_pyramid_pooling
        paths = [nn.Sequential(_pointwise3d(cin, self.half_channels),
                               CausalConv3dWeights(self.half_channels, self.half_channels, k))
                 for k in self.kernels]
        paths.append(_pointwise3d(cin, self.half_channels))
        self.convolution_paths = nn.ModuleList(paths)
        agg_in = len(paths) * self.half_channels
        self.reduction_channels = 0
        if use_pyramid_pooling:
            assert pool_sizes is not None, 'pool_sizes required for pyramid pooling'
            self.reduction_channels = cin // 3
            self.pyramid_pooling = PyramidPoolWeights(cin, self.reduction_channels, pool_sizes)
            agg_in += len(pool_sizes) * self.reduction_channels
        self.aggregation = nn.Sequential(_pointwise3d(agg_in, self.out_channels))
        if self.out_channels != cin:
            self.projection = nn.Sequential(nn.Conv3d(cin, self.out_channels, 1, bias=False),
                                            nn.BatchNorm3d(self.out_channels))
        else:
            self.projection = None


class Bottleneck3DWeights(nn.Module):
    def __init__(self, cin, cout=None, kernel_size=(2, 3, 3)):
        super().__init__()
        mid = cin // 2
        cout = cout or cin
        self.layers = _named(conv_down_project=_pointwise3d(cin, mid),
                             conv=CausalConv3dWeights(mid, mid, kernel_size),
                             conv_up_project=_pointwise3d(mid, cout))
        if cout != cin:
            self.projection = nn.Sequential(nn.Conv3d(cin, cout, 1, bias=False), nn.BatchNorm3d(cout))
        else:
            self.projection = None


class TemporalModelWeights(nn.Module):
    def __init__(self, in_channels, receptive_field, input_shape, start_out_channels=64,
                 extra_in_channels=0, n_spatial_layers_between_temporal_layers=0, use_pyramid_pooling=True):
        super().__init__()
        self.receptive_field = receptive_field
        h, w = input_shape
        stages = []
        cin, cout = in_channels, start_out_channels
        for _ in range(receptive_field - 1):
            stages.append(TemporalBlockWeights(cin, cout, use_pyramid_pooling=use_pyramid_pooling,
                                               pool_sizes=[(2, h, w)] if use_pyramid_pooling else None))
            stages.extend(Bottleneck3DWeights(cout, cout, kernel_size=(1, 3, 3))
                          for _ in range(n_spatial_layers_between_temporal_layers))
            cin = cout
            cout += extra_in_channels
        self.out_channels = cin
        self.model = nn.Sequential(*stages)


class TemporalIdentity(nn.Module):
    def __init__(self, in_channels, receptive_field):
        super().__init__()
        self.receptive_field = receptive_field
        self.out_channels = in_channels


# ----------------------------------------------------------------------------------------------
# future prediction
# ----------------------------------------------------------------------------------------------
class SpatialGRUWeights(nn.Module):
    def __init__(self, input_size, hidden_size, gru_bias_init=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        self.conv_update = nn.Conv2d(input_size + hidden_size, hidden_size, 3, bias=True, padding=1)
        self.conv_reset = nn.Conv2d(input_size + hidden_size, hidden_size, 3, bias=True, padding=1)
        self.conv_state_tilde = ConvNormAct2d(input_size + hidden_size, hidden_size, 3)


class FuturePredictionWeights(nn.Module):
    def __init__(self, in_channels, latent_dim, n_gru_blocks=3, n_res_layers=3):
        super().__init__()
        self.n_gru_blocks = n_gru_blocks
        self.spatial_grus = nn.ModuleList(
            SpatialGRUWeights(latent_dim if i == 0 else in_channels, in_channels) for i in range(n_gru_blocks))
        self.res_blocks = nn.ModuleList(
            nn.Sequential(*[ResidualBottleneck(in_channels) for _ in range(n_res_layers)])
            for _ in range(n_gru_blocks))


# ----------------------------------------------------------------------------------------------
# distributions
# ----------------------------------------------------------------------------------------------
class DistributionEncoderWeights(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.model = nn.Sequential(*[ResidualBottleneck(cin if i == 0 else cout, cout, downsample=True)
                                     for i in range(4)])


class DistributionWeights(nn.Module):
    def __init__(self, in_channels, latent_dim, min_log_sigma, max_log_sigma):
        super().__init__()
        self.compress_dim = in_channels // 2
        self.latent_dim = latent_dim
        self.min_log_sigma, self.max_log_sigma = min_log_sigma, max_log_sigma
        self.encoder = DistributionEncoderWeights(in_channels, self.compress_dim)
        self.last_conv = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                       nn.Conv2d(self.compress_dim, 2 * latent_dim, kernel_size=1))


# ----------------------------------------------------------------------------------------------
# decoder (resnet18 layers 1-3 restated; torchvision is not importable offline)
# ----------------------------------------------------------------------------------------------
class BasicBlockWeights(nn.Module):
    expansion = 1

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


def resnet18_stages(zero_init_residual=True):
    """`(bn1, relu, layer1, layer2, layer3)` with torchvision's resnet18 shapes and initialisation."""
    layers = []
    cin = 64
    for cout, stride in ((64, 1), (128, 2), (256, 2)):
        layers.append(nn.Sequential(BasicBlockWeights(cin, cout, stride), BasicBlockWeights(cout, cout, 1)))
        cin = cout
    bn1 = nn.BatchNorm2d(64)
    for module in [bn1, *layers]:
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
    if zero_init_residual:
        for layer in layers:
            for block in layer:
                nn.init.constant_(block.bn2.weight, 0)
    return bn1, nn.ReLU(inplace=True), layers[0], layers[1], layers[2]


def _head(cin, cout, sigmoid=False):
    parts = [nn.Conv2d(cin, cin, 3, padding=1, bias=False), nn.BatchNorm2d(cin), nn.ReLU(inplace=True),
             nn.Conv2d(cin, cout, 1, padding=0)]
    if sigmoid:
        parts.append(nn.Sigmoid())
    return nn.Sequential(*parts)


class DecoderWeights(nn.Module):
    def __init__(self, in_channels, n_classes, predict_future_flow):
        super().__init__()
        self.first_conv = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1, self.relu, self.layer1, self.layer2, self.layer3 = resnet18_stages(zero_init_residual=True)
        self.predict_future_flow = predict_future_flow
        self.in_channels, self.n_classes = in_channels, n_classes
        self.up3_skip = UpsampleAddWeights(256, 128)
        self.up2_skip = UpsampleAddWeights(128, 64)
        self.up1_skip = UpsampleAddWeights(64, in_channels)
        self.segmentation_head = _head(in_channels, n_classes)
        self.instance_offset_head = _head(in_channels, 2)
        self.instance_center_head = _head(in_channels, 1, sigmoid=True)
        if predict_future_flow:
            self.instance_future_head = _head(in_channels, 2)


def set_bn_momentum(model, momentum=0.1):
    """reference: fiery/utils/network.py:28-31"""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = momentum
