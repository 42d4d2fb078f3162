"""Pins the oracle against the reference's own code, executed here on the CPU (build container only)."""
import numpy as np
import pytest
import torch

from fiery_amd.synthetic import make_inputs, make_lifted_features
from oracle import bev_stack
from oracle import instance as oi
from oracle import lift_splat as ls
from tests.helpers import forward_case, randomise_weights, tiny_cfg

pytestmark = pytest.mark.needs_reference


@pytest.fixture(scope='module')
def ref():
    from oracle.ref_shims import load_reference
    return load_reference()


def test_state_dict_keys_and_shapes_match_reference(ref):
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    for preset in ('baseline.yml', 'literature/static_lss_setting.yml', 'literature/pon_setting.yml', 'lyft/baseline.yml'):
        cfg = get_preset_cfg(preset)
        theirs = ref.Fiery(cfg).state_dict()
        ours = Fiery(cfg).state_dict()
        assert list(theirs) == list(ours), preset
        for k in theirs:
            assert theirs[k].shape == ours[k].shape and theirs[k].dtype == ours[k].dtype, (preset, k)
        ref.Fiery(cfg).load_state_dict(ours, strict=True)


def test_geometry_and_pooling_equal_reference_bitwise(ref):
    cfg = tiny_cfg('baseline.yml', bev=16)
    torch.manual_seed(0)
    m = ref.Fiery(cfg).eval()
    _, K, E, _ = make_inputs(1, 2, 3, with_image=False)
    _, _, lifted = make_lifted_features(6, 8, m.depth_channels, (8, 12), seed=9)
    lifted = lifted.view(2, 3, 8, m.depth_channels, 8, 12)
    with torch.no_grad():
        geo_ref = m.get_geometry(K[0], E[0]).numpy()
        bev_ref = m.projection_to_birds_eye_view(lifted.permute(0, 1, 3, 4, 5, 2), torch.from_numpy(geo_ref)).numpy()
    res, start, dim = ls.bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
    geo = ls.get_geometry(m.frustum.numpy(), K[0].numpy(), E[0].numpy())
    assert np.array_equal(geo, geo_ref)
    for f in range(2):
        got = ls.voxel_pool_reference(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.array_equal(got, bev_ref[f])


def test_pooling_backward_equals_reference_autograd_bitwise(ref):
    """`VoxelsSumming.backward` + the autograd graph around it (fiery.py:233-271), run by the reference itself."""
    cfg = tiny_cfg('baseline.yml', bev=16)
    torch.manual_seed(0)
    m = ref.Fiery(cfg)
    _, K, E, _ = make_inputs(1, 2, 3, with_image=False)
    _, _, lifted = make_lifted_features(6, 8, m.depth_channels, (8, 12), seed=9)
    x = lifted.view(2, 3, 8, m.depth_channels, 8, 12).permute(0, 1, 3, 4, 5, 2).clone().requires_grad_(True)
    with torch.no_grad():
        geo = m.get_geometry(K[0], E[0])
    bev = m.projection_to_birds_eye_view(x, geo)
    grad_bev = torch.randn(bev.shape, generator=torch.Generator().manual_seed(5))
    bev.backward(grad_bev)
    res, start, dim = ls.bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
    for f in range(2):
        got = ls.voxel_pool_backward(grad_bev[f].numpy(), geo[f].numpy().reshape(-1, 3), res, start, dim)
        assert np.array_equal(got, x.grad[f].numpy().reshape(-1, 8))
    assert (x.grad != 0).any()


def test_warp_matches_reference(ref):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 4, 10, 12, generator=g)
    _, _, _, ego = make_inputs(2, 3, 1, with_image=False)
    ego[..., 1] = 0.3
    want = ref.geometry.cumulative_warp_features(x.clone(), ego, mode='bilinear', spatial_extent=(5.0, 6.0))
    got = bev_stack.cumulative_warp_features(x.clone(), ego, 'bilinear', (5.0, 6.0))
    assert torch.equal(got, want)


def test_product_warp_kernel_is_bitwise_the_reference(ref, sim):
    """The product's resampling kernel (its source on the CPU simulator) fed with `host_warp_transforms` against the
    REFERENCE's `cumulative_warp_features` itself, white noise at the baseline map size: equal bit for bit."""
    from fiery_amd.model import host_warp_transforms
    g = torch.Generator().manual_seed(7)
    B, S, C, H, W = 1, 3, 2, 200, 200
    x = torch.randn(B, S, C, H, W, generator=g)
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 2.5 + 0.5 * torch.rand(B, S, generator=g)
    ego[..., 1] = 0.1 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.02 * torch.randn(B, S, generator=g)
    want = ref.geometry.cumulative_warp_features(x.clone(), ego, mode='bilinear', spatial_extent=(50.0, 50.0))
    theta = host_warp_transforms(ego, (50.0, 50.0))
    out = torch.zeros(B * S, H, W, 8)
    sim.bev_warp_nchw_to_nhwc(x.view(B * S, C, H, W), theta.view(B * S, 6).contiguous(), [False, False, True], out, 8, H * W * 8)
    got = out[..., :C].permute(0, 3, 1, 2).reshape(B, S, C, H, W)
    assert torch.equal(got, want)


@pytest.mark.parametrize('preset,B,labels', [('baseline.yml', 2, True), ('literature/static_lss_setting.yml', 1, False),
                                             ('literature/pon_setting.yml', 1, False), ('lyft/baseline.yml', 1, True),
                                             ('literature/fishing_setting.yml', 1, False),
                                             ('temporal_single_timeframe.yml', 1, False), ('single_timeframe.yml', 1, False)])
def test_hot_path_equals_reference_forward(ref, preset, B, labels):
    from tests.golden.make_golden import run_reference_from_lifted
    cfg = tiny_cfg(preset)
    torch.manual_seed(0)
    m = ref.Fiery(cfg).eval()
    sd = randomise_weights(m)
    lifted, K, E, ego, lab, noise = forward_case(cfg, m.receptive_field, m.n_future, m.depth_channels, m.bev_size, B, 2,
                                                 with_labels=labels, with_noise=labels)
    want = run_reference_from_lifted(m, lifted, K, E, ego, lab, noise)
    with torch.no_grad():
        got = bev_stack.bev_hot_path(sd, cfg, lifted, K, E, ego, noise)
    for k, v in got.items():
        if v is None:
            continue
        assert torch.allclose(v, want[k], rtol=1e-5, atol=1e-5), k


def _instance_case(seed, H=40, W=56, n_blobs=7, all_foreground=False):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float), torch.arange(W, dtype=torch.float), indexing='ij')
    center = torch.zeros(H, W)
    cy, cx = torch.rand(n_blobs, generator=g) * H, torch.rand(n_blobs, generator=g) * W
    for k in range(n_blobs):
        center = torch.maximum(center, torch.exp(-((yy - cy[k]) ** 2 + (xx - cx[k]) ** 2) / 9.0))
    center = center + 0.01 * torch.rand(H, W, generator=g)
    nearest = torch.stack([(yy - cy[k]) ** 2 + (xx - cx[k]) ** 2 for k in range(n_blobs)]).argmin(0)
    offset = torch.stack([cy[nearest] - yy, cx[nearest] - xx]) + 0.3 * torch.randn(2, H, W, generator=g)
    fg = torch.ones(H, W, dtype=torch.bool) if all_foreground else center > 0.05
    return center, offset, fg


@pytest.mark.parametrize('seed,kw', [(0, {}), (1, dict(n_blobs=1)), (2, dict(all_foreground=True)), (3, dict(n_blobs=140, H=64, W=64))])
def test_instance_segmentation_oracle_equals_reference(ref, seed, kw):
    """fiery/utils/instance.py:116-144 run by the reference itself (more than 100 centres, no background pixel, ...)."""
    center, offset, fg = _instance_case(seed, **kw)
    want_seg, want_centers = ref.instance.get_instance_segmentation_and_centers(center.clone(), offset.clone(), fg.clone())
    got_seg, got_centers = oi.instance_segmentation_and_centers(center, offset, fg)
    assert torch.equal(got_seg, want_seg) and torch.equal(got_centers, want_centers)
    assert got_seg.max() > 0
    empty_seg, empty_centers = oi.instance_segmentation_and_centers(torch.zeros_like(center), offset, fg)
    ref_seg, ref_centers = ref.instance.get_instance_segmentation_and_centers(torch.zeros_like(center), offset, fg)
    assert torch.equal(empty_seg, ref_seg) and len(empty_centers) == len(ref_centers) == 0


def _video_case(seed, B=2, T=3, H=48, W=64, n_obj=6):
    """Model-output-like tensors for a few moving blobs: segmentation logits, centerness, offsets, flow."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float), torch.arange(W, dtype=torch.float), indexing='ij')
    out = dict(segmentation=torch.zeros(B, T, 2, H, W), instance_center=torch.zeros(B, T, 1, H, W),
               instance_offset=torch.zeros(B, T, 2, H, W), instance_flow=torch.zeros(B, T, 2, H, W))
    for b in range(B):
        pos = torch.stack([torch.rand(n_obj, generator=g) * (H - 12) + 6, torch.rand(n_obj, generator=g) * (W - 12) + 6], 1)
        vel = torch.randn(n_obj, 2, generator=g) * 1.5
        alive = torch.ones(n_obj, T, dtype=torch.bool)
        alive[0, T - 1] = False                                       # one object disappears,
        alive[1, 0] = False                                           # one appears later
        for t in range(T):
            p = pos + vel * t
            d2 = torch.stack([(yy - p[k, 0]) ** 2 + (xx - p[k, 1]) ** 2 + (0 if alive[k, t] else 1e6) for k in range(n_obj)])
            nearest = d2.argmin(0)
            center = torch.exp(-d2.min(0).values / 9.0)
            out['instance_center'][b, t, 0] = center + 0.01 * torch.rand(H, W, generator=g)
            out['instance_offset'][b, t, 0] = p[nearest, 0] - yy
            out['instance_offset'][b, t, 1] = p[nearest, 1] - xx
            out['instance_flow'][b, t, 0] = vel[nearest, 0]
            out['instance_flow'][b, t, 1] = vel[nearest, 1]
            fg = center > 0.2
            out['segmentation'][b, t, 1] = fg.float() * 4 - 2
    return out


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_instance_trajectories_equal_the_reference(ref, sim, seed):
    """`predict_instance_segmentation_and_trajectories` (evaluate.py:62): per-frame segmentation on the kernel sources
    (simulator), temporal matching restated on the host - against the reference's own function, id for id."""
    from fiery_amd import instance as hip_instance
    out = _video_case(seed)
    want = ref.instance.predict_instance_segmentation_and_trajectories({k: v.clone() for k, v in out.items()})
    got = hip_instance.predict_instance_segmentation_and_trajectories(out, lib=sim)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert torch.equal(got, want)
    assert want.max() >= 4
    no_flow = dict(out, instance_flow=None)
    want = ref.instance.predict_instance_segmentation_and_trajectories({k: (None if v is None else v.clone()) for k, v in no_flow.items()})
    got = hip_instance.predict_instance_segmentation_and_trajectories(dict(no_flow), lib=sim)
    assert torch.equal(got, want)


# ---- the step after the path, continued: metrics and the visualiser's trajectories ------------------------------------
def test_matched_centers_equal_the_reference(ref, sim):
    """`compute_matched_centers=True` (fiery/utils/instance.py:308-328; batch 1): same instances, same trajectories."""
    from fiery_amd import instance as hip_instance
    out = {k: v[:1] for k, v in _video_case(3, B=1, T=4).items()}
    want_seg, want = ref.instance.predict_instance_segmentation_and_trajectories({k: v.clone() for k, v in out.items()},
                                                                                  compute_matched_centers=True)
    got_seg, got = hip_instance.predict_instance_segmentation_and_trajectories(out, compute_matched_centers=True, lib=sim)
    assert torch.equal(got_seg, want_seg)
    assert sorted(got) == sorted(want) and len(want) >= 4
    for key in want:
        assert got[key].shape == want[key].shape
        assert np.allclose(got[key], want[key], rtol=0, atol=1e-4), key


def _instance_video(seed, B=2, T=4, H=40, W=48, n_obj=7, drift=True):
    """Ground-truth-like and prediction-like instance id videos: blobs that move; the prediction loses one object, invents
    one, swaps an id halfway through (a temporal inconsistency) and is a little smaller than the truth."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float), torch.arange(W, dtype=torch.float), indexing='ij')
    gt = torch.zeros(B, T, H, W, dtype=torch.long)
    pred = torch.zeros(B, T, H, W, dtype=torch.long)
    for b in range(B):
        pos = torch.stack([torch.rand(n_obj, generator=g) * (H - 10) + 5, torch.rand(n_obj, generator=g) * (W - 10) + 5], 1)
        vel = torch.randn(n_obj, 2, generator=g)
        for t in range(T):
            p = pos + vel * t
            for k in range(n_obj):
                d2 = (yy - p[k, 0]) ** 2 + (xx - p[k, 1]) ** 2
                gt[b, t][d2 < 12] = k + 1
                if k == 0:
                    continue                                          # never detected
                pid = k + 1
                if drift and k == 2 and t >= 2:
                    pid = n_obj + 5                                    # its id changes: not temporally consistent
                pred[b, t][d2 < (9 if k % 2 else 12)] = pid
            pred[b, t][(yy - 3) ** 2 + (xx - 3) ** 2 < 6] = n_obj + 9  # a false alarm
    return pred, gt


@pytest.mark.parametrize('seed,consistent', [(0, True), (1, True), (2, False)])
def test_panoptic_metric_equals_the_reference(ref, seed, consistent):
    from fiery_amd.metrics import PanopticMetric
    pred, gt = _instance_video(seed)
    want_metric = ref.metrics.PanopticMetric(n_classes=2, temporally_consistent=consistent)
    got_metric = PanopticMetric(n_classes=2, temporally_consistent=consistent)
    for _ in range(2):                                                 # states accumulate over updates
        want_metric(pred, gt)
        got_metric(pred, gt)
    want, got = want_metric.compute(), got_metric.compute()
    assert set(want) == set(got)
    for key in want:
        assert torch.allclose(got[key], want[key], rtol=1e-6, atol=1e-6), (key, got[key], want[key])
    assert want['pq'][1] > 0.1 and want_metric.false_positive[1] > 0 and want_metric.false_negative[1] > 0
    got_metric.reset()
    assert float(got_metric.iou.sum()) == 0.0


@pytest.mark.parametrize('ignore_index,reduction', [(None, 'none'), (0, 'none'), (None, 'elementwise_mean')])
def test_intersection_over_union_equals_the_reference(ref, ignore_index, reduction):
    from fiery_amd.metrics import IntersectionOverUnion
    g = torch.Generator().manual_seed(4)
    n_classes = 4
    want_metric = ref.metrics.IntersectionOverUnion(n_classes, ignore_index=ignore_index, absent_score=0.5, reduction=reduction)
    got_metric = IntersectionOverUnion(n_classes, ignore_index=ignore_index, absent_score=0.5, reduction=reduction)
    for _ in range(3):
        target = torch.randint(0, 3, (2, 5, 1, 20, 24), generator=g)   # class 3 never occurs: the absent score applies
        pred = torch.where(torch.rand(target.shape, generator=g) < 0.7, target, torch.randint(0, 3, target.shape, generator=g))
        want_metric(pred, target)
        got_metric(pred, target)
    assert torch.allclose(got_metric.compute(), want_metric.compute(), rtol=1e-6, atol=1e-7)


# ---- the step before the path on the training side: label warping (trainer.py:133-191) --------------------------------
def _label_batch(seed, B=2, S=7, H=40, W=48):
    """A dataset-like batch: label videos in each frame's own ego frame, plus ego-motion."""
    g = torch.Generator().manual_seed(seed)
    inst = torch.zeros(B, S, H, W, dtype=torch.long)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float), torch.arange(W, dtype=torch.float), indexing='ij')
    for b in range(B):
        for k in range(5):
            cy, cx = torch.rand(2, generator=g) * torch.tensor([H - 10.0, W - 10.0]) + 5
            for t in range(S):
                inst[b, t][((yy - cy - 0.7 * t) ** 2 + (xx - cx + 0.4 * t) ** 2) < 14] = k + 1
    ego = torch.zeros(B, S, 6)
    ego[..., 0] = 0.6 + 0.3 * torch.rand(B, S, generator=g)
    ego[..., 1] = 0.1 * torch.randn(B, S, generator=g)
    ego[..., 5] = 0.03 * torch.randn(B, S, generator=g)
    return dict(segmentation=(inst > 0).long().unsqueeze(2), instance=inst, centerness=torch.rand(B, S, 1, H, W, generator=g),
                offset=torch.randn(B, S, 2, H, W, generator=g), flow=torch.randn(B, S, 2, H, W, generator=g), future_egomotion=ego)


def test_reverse_warp_oracle_is_bitwise_the_reference(ref):
    batch = _label_batch(0)
    extent = (10.0, 12.0)
    for key in ('centerness', 'offset'):
        want = ref.geometry.cumulative_warp_features_reverse(batch[key][:, 2:], batch['future_egomotion'][:, 2:], mode='nearest',
                                                             spatial_extent=extent)
        got = bev_stack.cumulative_warp_features_reverse(batch[key][:, 2:], batch['future_egomotion'][:, 2:], 'nearest', extent)
        assert torch.equal(got, want)


def test_prepare_future_labels_equals_the_reference(ref, sim):
    """`fiery_amd.labels.prepare_future_labels` on the kernel sources (simulator) against `FieryTrainer.prepare_future_labels`
    run on the reference's own geometry functions.  Nearest-neighbour sampling of an id map is discontinuous: a sampling
    position that ATen puts a hair on one side of a pixel boundary and the kernel on the other gives a different id, so
    equality is asked of all but a handful of pixels."""
    import types
    from fiery_amd import labels as hip_labels
    batch = _label_batch(1)
    extent = (10.0, 12.0)
    rf = 3
    # the reference method only needs `self.model.receptive_field`, `self.spatial_extent`, `self.cfg.INSTANCE_FLOW.ENABLED`;
    # its module (fiery/trainer.py) imports pytorch_lightning and the dataset code, so the method body is exercised through
    # the geometry function it calls, step by step as trainer.py:143-187 does
    warp = lambda t: ref.geometry.cumulative_warp_features_reverse(t[:, rf - 1:], batch['future_egomotion'][:, rf - 1:], mode='nearest',
                                                                   spatial_extent=extent)
    want = dict(segmentation=warp(batch['segmentation'].float()).long().contiguous(),
                instance=warp(batch['instance'].float().unsqueeze(2)).long().contiguous()[:, :, 0],
                centerness=warp(batch['centerness']).contiguous(), offset=warp(batch['offset']).contiguous(),
                flow=warp(batch['flow']).contiguous())
    want_inputs = torch.cat([want['segmentation'], want['centerness'], want['offset'], want['flow']], dim=2)
    got, got_inputs = hip_labels.prepare_future_labels(batch, rf, extent, instance_flow_enabled=True, lib=sim)
    assert set(got) == set(want)
    for key in want:
        assert got[key].shape == want[key].shape and got[key].dtype == want[key].dtype, key
        differing = (got[key] != want[key]).reshape(got[key].shape[0], -1).any(dim=0).float().mean().item() if got[key].dim() == 4 \
            else (got[key] != want[key]).float().mean().item()
        assert differing < 2e-3, (key, differing)
        assert torch.equal(got[key][:, 0], want[key][:, 0])            # the present frame is never resampled
    assert got_inputs.shape == want_inputs.shape == (2, 5, 6, 40, 48)
    assert (got['instance'] != batch['instance'][:, rf - 1:]).float().mean() > 0.01     # the warp does move things


def _instance_label_case(seed, T=5, H=40, W=56, n_obj=7):
    """Moving blobs with appearing / disappearing objects + ego-motion with translation and yaw."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(T, H, W, dtype=torch.int64)
    centres = torch.rand(n_obj, 2, generator=g) * torch.tensor([H - 10.0, W - 10.0]) + 5.0
    speed = torch.randn(n_obj, 2, generator=g) * 1.5
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    for t in range(T):
        for k in range(n_obj):
            if (k + t + seed) % 5 == 0:                    # the object is missing in this frame
                continue
            c = centres[k] + speed[k] * t
            ids[t][((yy - c[0]).abs() < 2.5 + k % 3) & ((xx - c[1]).abs() < 3.5)] = k + 1
    ego = torch.zeros(T, 6)
    ego[:, 0] = 1.0 + torch.rand(T, generator=g)
    ego[:, 1] = 0.3 * torch.randn(T, generator=g)
    ego[:, 5] = 0.05 * torch.randn(T, generator=g)
    return ids, ego, n_obj


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_instance_label_oracle_and_kernels_equal_the_reference(ref, sim, seed):
    """`convert_instance_mask_to_center_and_offset_label` (fiery/utils/instance.py:12-77): the oracle and the product on the
    simulated kernels against the reference function - offsets and displacements exactly, the heat map to an ulp of exp."""
    from fiery_amd.labels import convert_instance_mask_to_center_and_offset_label
    from oracle.labels import instance_labels
    ids, ego, n = _instance_label_case(seed)
    extent = (20.0, 28.0)
    want = ref.instance.convert_instance_mask_to_center_and_offset_label(ids, ego, n, ignore_index=255, subtract_egomotion=True,
                                                                         spatial_extent=extent)
    for got in (instance_labels(ids, ego, n, 255, 3, extent),
                convert_instance_mask_to_center_and_offset_label(ids, ego, n, ignore_index=255, spatial_extent=extent, lib=sim, device='cpu')):
        assert torch.allclose(got[0], want[0], rtol=0, atol=2e-7)
        assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    assert (want[2] != 255).any() and (want[1] != 255).any()


def test_update_intrinsics_equals_the_reference(ref):
    from fiery_amd.images import update_intrinsics
    K = torch.tensor([[1266.417, 0.0, 816.267], [0.0, 1266.417, 491.507], [0.0, 0.0, 1.0]])
    for args in ((46, 0, 0.3, 0.3), (0.0, 12.0, 0.25, 0.3), (0, 0, 1.0, 1.0)):
        assert torch.equal(update_intrinsics(K, *args), ref.geometry.update_intrinsics(K, *args))
