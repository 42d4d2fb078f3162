"""The two third-party networks exist twice in this repo, written independently: the product's restatement
(`fiery_amd/backbone.py`, `fiery_amd/modules.py` - what the HIP engine reads its weights from and what the torch statement
of the trunk runs) and the checker's (`oracle/third_party.py` - what the reference's own code is run on to make fixtures
and to pin the oracle).  They must agree: same `state_dict`, same numbers for the same weights."""
import torch

from fiery_amd import backbone, modules
from oracle import third_party


def _same_state_dict_layout(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    assert all(sa[k].shape == sb[k].shape and sa[k].dtype == sb[k].dtype for k in sa)


def test_efficientnet_b4_trunks_agree():
    torch.manual_seed(0)
    ours = backbone.EfficientNet.from_pretrained('efficientnet-b4').eval()
    theirs = third_party.EfficientNet.from_pretrained('efficientnet-b4').eval()
    _same_state_dict_layout(ours, theirs)
    assert len(ours._blocks) == len(theirs._blocks) == 32
    # non-trivial weights and BatchNorm statistics, identical on both sides
    sd = ours.state_dict()
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():
        if k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith('running_mean'):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dtype.is_floating_point and v.dim() >= 1:
            sd[k] = v + 0.05 * torch.randn(v.shape, generator=g)
    ours.load_state_dict(sd)
    theirs.load_state_dict(sd)
    x = torch.randn(2, 3, 64, 96, generator=g)
    with torch.no_grad():
        a = ours._swish(ours._bn0(ours._conv_stem(x)))
        b = theirs._swish(theirs._bn0(theirs._conv_stem(x)))
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        for idx in range(22):                     # the blocks the lift head keeps (fiery/models/encoder.py:46-50, 79-80)
            assert tuple(ours._blocks[idx]._depthwise_conv.static_padding.padding) == \
                tuple(theirs._blocks[idx]._depthwise_conv.padding), idx
            a = ours._blocks[idx](a, drop_connect_rate=0.2 * idx / 32)
            b = theirs._blocks[idx](b, drop_connect_rate=0.2 * idx / 32)
            assert a.shape == b.shape
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), (idx, (a - b).abs().max())


def test_resnet18_stages_agree_in_layout_and_initialisation():
    torch.manual_seed(0)
    bn1, relu, l1, l2, l3 = modules.resnet18_stages(zero_init_residual=True)
    net = third_party.resnet18(pretrained=False, zero_init_residual=True)
    _same_state_dict_layout(bn1, net.bn1)
    for ours, theirs in ((l1, net.layer1), (l2, net.layer2), (l3, net.layer3)):
        _same_state_dict_layout(ours, theirs)
        for blk_o, blk_t in zip(ours, theirs):
            assert blk_o.stride == blk_t.stride
            assert float(blk_o.bn2.weight.detach().abs().max()) == float(blk_t.bn2.weight.detach().abs().max()) == 0.0       # zero_init_residual
    # the block's arithmetic: the checker's module against the oracle's functional statement of it
    from oracle.bev_stack import Weights, _basic_block
    g = torch.Generator().manual_seed(2)
    blk = net.layer2[0].eval()
    sd = blk.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            sd[k] = (0.5 + torch.rand(v.shape, generator=g)) if k.endswith('running_var') else 0.2 * torch.randn(v.shape, generator=g)
    blk.load_state_dict(sd)
    x = torch.randn(2, 64, 12, 10, generator=g)
    with torch.no_grad():
        assert torch.allclose(blk(x), _basic_block(x, Weights(blk.state_dict())), rtol=1e-5, atol=1e-6)


def test_efficientnet_parameter_counts_are_the_published_ones_and_the_manifest_holds():
    """Two external anchors for the absent `efficientnet-pytorch==0.7.0` (fiery/models/encoder.py:16): (1) the whole network's
    parameter count - 5,288,548 for b0 and 19,341,616 for b4, the figures published with the package and the paper's tables -
    which fixes stem / stage widths, repeats, kernel sizes, expansion and squeeze ratios, head and classifier together;
    (2) `tests/golden/efficientnet_manifest.json`: every state_dict key with its shape (706 keys for b4), the list a released
    FIERY checkpoint's `encoder.backbone.*` entries are checked against before loading.  Both restatements -
    the product's and the checker's - must agree with both."""
    import json
    import os
    from fiery_amd.backbone import EfficientNet as Product
    from oracle.third_party import EfficientNet as Checker
    manifest = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'efficientnet_manifest.json')))
    published = {'efficientnet-b0': 5288548, 'efficientnet-b4': 19341616}
    for name, count in published.items():
        for cls in (Product, Checker):
            net = cls.from_pretrained(name)
            assert sum(p.numel() for p in net.parameters()) == count, (name, cls.__module__)
            assert {k: list(v.shape) for k, v in net.state_dict().items()} == manifest[name], (name, cls.__module__)
