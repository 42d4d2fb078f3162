"""Image encoder and lift head (the stage *before* the BEV hot path).

Interface and `state_dict` names follow the reference's `Encoder`
(reference: fiery/models/encoder.py:7-104): `backbone.*`, `upsampling_layer.conv.{0,1,3,4}.*`,
`depth_layer.*`.  On the GPU the trunk and the lift head behind it (x2 bilinear of the coarse
level, concat, two 3x3 conv+BN+ReLU, the 1x1 depth layer) run on the HIP engine in inference (`engine.BevEngine`: `_MBConv`,
`lift_head`) and on `train_graph`'s operators in training; the modules below hold the parameters under the reference's names; its two results - depth logits and context features - feed the fused HIP
lift-splat kernel directly, so the (n, C, D, h, w) outer product never has to exist unless a caller asks for it via
`forward()`.  The torch statement of the head below is what autograd and `encoder_forward` use.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import UpsampleConcatWeights

try:                                    # the real package, when a deployment has it
    from efficientnet_pytorch import EfficientNet
except ImportError:                     # offline: architectural restatement with the same key names
    from .backbone import EfficientNet

# ((deep endpoint channels, shallow endpoint channels), fused channels) per (downsample, version)
_UPSAMPLING_PLAN = {
    (16, 'b0'): ((320, 112), 512), (16, 'b4'): ((448, 160), 512),
    (8, 'b0'): ((112, 40), 128), (8, 'b4'): ((160, 56), 128),
}
# index of the last trunk block that is kept when downsampling by 8 (reference: encoder.py:43-47)
_LAST_BLOCK_DS8 = {'b0': 10, 'b4': 21}


class Encoder(nn.Module):
    def __init__(self, cfg, D):
        super().__init__()
        self.D = D
        self.C = cfg.OUT_CHANNELS
        self.use_depth_distribution = cfg.USE_DEPTH_DISTRIBUTION
        self.downsample = cfg.DOWNSAMPLE
        self.version = cfg.NAME.split('-')[1]
        if (self.downsample, self.version) not in _UPSAMPLING_PLAN:
            raise ValueError(f'Downsample factor {self.downsample} / trunk {cfg.NAME} not handled.')

        self.backbone = EfficientNet.from_pretrained(cfg.NAME)
        self._drop_unused_trunk_layers()

        (self.c_deep, self.c_shallow), cout = _UPSAMPLING_PLAN[(self.downsample, self.version)]
        self.upsampling_layer = UpsampleConcatWeights(self.c_deep + self.c_shallow, cout)
        head_out = self.C + self.D if self.use_depth_distribution else self.C
        self.depth_layer = nn.Conv2d(cout, head_out, kernel_size=1, padding=0)

    def _drop_unused_trunk_layers(self):
        if self.downsample == 8:
            last = _LAST_BLOCK_DS8[self.version]
            for idx in reversed(range(last + 1, len(self.backbone._blocks))):
                del self.backbone._blocks[idx]
        for name in ('_conv_head', '_bn1', '_avg_pooling', '_dropout', '_fc'):
            delattr(self.backbone, name)

    def trunk_endpoints(self, x):
        """The image trunk: -> (deep, shallow) pyramid levels, the coarse one at half the resolution of the fine one
        (reference: encoder.py:58-86)."""
        trunk = self.backbone
        endpoints = []
        x = trunk._swish(trunk._bn0(trunk._conv_stem(x)))
        previous = x
        n_blocks = len(trunk._blocks)
        for idx, block in enumerate(trunk._blocks):
            rate = trunk._global_params.drop_connect_rate
            if rate:
                rate *= float(idx) / n_blocks
            x = block(x, drop_connect_rate=rate)
            if previous.size(2) > x.size(2):
                endpoints.append(previous)
            previous = x
            if self.downsample == 8 and idx == _LAST_BLOCK_DS8[self.version]:
                break
        endpoints.append(x)
        # downsample 16 -> reductions 5 and 4; downsample 8 -> reductions 4 and 3
        return (endpoints[4], endpoints[3]) if self.downsample == 16 else (endpoints[3], endpoints[2])

    def get_features(self, x):
        """Trunk -> two pyramid levels -> upsample-concat-conv (reference: encoder.py:58-91)."""
        deep, shallow = self.trunk_endpoints(x)
        deep = F.interpolate(deep, scale_factor=2, mode='bilinear', align_corners=False)
        x = torch.cat([shallow, deep], dim=1)
        conv = self.upsampling_layer.conv
        x = F.relu(conv[1](conv[0](x)))
        return F.relu(conv[4](conv[3](x)))

    def lift_head(self, x):
        """-> (depth_logits (n, D, h, w) or None, features (n, C, h, w))."""
        x = self.depth_layer(self.get_features(x))
        if self.use_depth_distribution:
            return x[:, :self.D], x[:, self.D:(self.D + self.C)]
        return None, x

    def forward(self, x):
        """(n, 3, H, W) -> (n, C, D, h, w), as the reference's encoder returns (encoder.py:93-104)."""
        depth_logits, features = self.lift_head(x)
        if depth_logits is None:
            return features.unsqueeze(2).repeat(1, 1, self.D, 1, 1)
        return depth_logits.softmax(dim=1).unsqueeze(1) * features.unsqueeze(2)
