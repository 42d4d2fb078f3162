"""Seeded synthetic inputs with the tensor contract of the reference's dataset
(reference: fiery/data.py:345-367 - image (S,n,3,H,W), intrinsics (S,n,3,3),
extrinsics (S,n,4,4) camera->ego, future_egomotion (S,6)).

No dataset is available offline, so benchmarks, smoke tests and golden fixtures all
draw from here.  Everything is generated on the CPU from explicit generators so that
the same seed gives the same tensors on every machine with this torch build.
"""
import math

import torch

# nuScenes-like rig after the reference's resize 0.3 / top-crop 46 (fiery/config.py:61-62):
# focal 1266 px * 0.3, principal point (816, 491) * 0.3 - (0, 46).
_FOCAL = 380.0
_CX = 245.0
_CY = 101.0
_YAWS_DEG = (55.0, 0.0, -55.0, 110.0, 180.0, -110.0, 90.0, -90.0)


def _rot_z(yaw):
    c, s = math.cos(yaw), math.sin(yaw)
    return torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)


def camera_rig(n_cameras, jitter=True, dtype=torch.float32):
    """Intrinsics (n,3,3) and camera->ego extrinsics (n,4,4) of a surround rig.

    `jitter=False` gives axis-aligned cameras whose frustum points fall *exactly* on voxel
    edges (integer depths, round offsets): the boundary-stress case for the index path.
    `jitter=True` adds small non-round offsets, the realistic case.
    """
    if n_cameras > len(_YAWS_DEG):
        raise ValueError(f'at most {len(_YAWS_DEG)} synthetic cameras')
    cam_to_ego_axes = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]], dtype=torch.float64)
    intrinsics = torch.zeros(n_cameras, 3, 3, dtype=torch.float64)
    extrinsics = torch.zeros(n_cameras, 4, 4, dtype=torch.float64)
    for i in range(n_cameras):
        yaw = math.radians(_YAWS_DEG[i] + (0.37 + 0.11 * i if jitter else 0.0))
        pitch = math.radians(0.4 - 0.15 * i) if jitter else 0.0
        rz = _rot_z(yaw)
        cp, sp = math.cos(pitch), math.sin(pitch)
        ry = torch.tensor([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]], dtype=torch.float64)
        rotation = rz @ ry @ cam_to_ego_axes
        offset = torch.tensor([1.5, 0.0, 1.5], dtype=torch.float64)
        if jitter:
            offset = offset + torch.tensor([0.013 + 0.002 * i, -0.007 + 0.003 * i, 0.004], dtype=torch.float64)
        extrinsics[i, :3, :3] = rotation
        extrinsics[i, :3, 3] = rz @ offset
        extrinsics[i, 3, 3] = 1.0
        f = _FOCAL + (0.731 * i if jitter else 0.0)
        intrinsics[i] = torch.tensor([[f, 0.0, _CX + (0.21 * i if jitter else 0.0)],
                                      [0.0, f + (0.05 if jitter else 0.0), _CY - (0.13 * i if jitter else 0.0)],
                                      [0.0, 0.0, 1.0]], dtype=torch.float64)
    return intrinsics.to(dtype), extrinsics.to(dtype)


def make_inputs(batch, n_frames, n_cameras, image_hw=(224, 480), seed=0, jitter=True, with_image=True):
    """Model inputs `(image, intrinsics, extrinsics, future_egomotion)` for `Fiery.forward`."""
    gen = torch.Generator().manual_seed(seed)
    intr, extr = camera_rig(n_cameras, jitter=jitter)
    intrinsics = intr.expand(batch, n_frames, n_cameras, 3, 3).contiguous()
    extrinsics = extr.expand(batch, n_frames, n_cameras, 4, 4).contiguous()
    ego = torch.zeros(batch, n_frames, 6)
    ego[..., 0] = 2.5 + 0.5 * torch.rand(batch, n_frames, generator=gen)      # forward metres / step
    ego[..., 5] = 0.02 * torch.randn(batch, n_frames, generator=gen)            # yaw radians / step
    image = None
    if with_image:
        image = torch.randn(batch, n_frames, n_cameras, 3, image_hw[0], image_hw[1], generator=gen)
    return image, intrinsics, extrinsics, ego


def make_lifted_features(n_images, channels, depth, feat_hw, seed=1, materialise=True):
    """Stand-ins for the image encoder's outputs, drawn zero-mean so the reference's
    prefix-sum pooling stays inside its own noise floor.

    Returns `(depth_logits (n,D,h,w), features (n,C,h,w), lifted (n,C,D,h,w) or None)` where
    `lifted = softmax_D(depth_logits) (x) features` is what the reference's encoder returns
    (reference: fiery/models/encoder.py:96-102).
    """
    gen = torch.Generator().manual_seed(seed)
    fh, fw = feat_hw
    depth_logits = torch.randn(n_images, depth, fh, fw, generator=gen)
    features = torch.randn(n_images, channels, fh, fw, generator=gen)
    lifted = None
    if materialise:
        lifted = depth_logits.softmax(dim=1).unsqueeze(1) * features.unsqueeze(2)
    return depth_logits, features, lifted


def randomise_weights(model, seed=2):
    """Non-trivial values everywhere, BatchNorm statistics included (fresh BN is the identity: a weak test).
    Deterministic per key name, so any module tree with the same state_dict keys gets the same values."""
    sd = model.state_dict()
    new = {}
    for i, key in enumerate(sorted(sd)):
        t = sd[key]
        if key in ('frustum', 'bev_resolution', 'bev_start_position', 'bev_dimension') or key.endswith('num_batches_tracked'):
            new[key] = t
            continue
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        if key.endswith('running_var'):
            v = 0.5 + torch.rand(t.shape, generator=g)
        elif key.endswith('running_mean'):
            v = 0.2 * torch.randn(t.shape, generator=g)
        elif t.dim() == 0:                                     # (the trainer's uncertainty weights, fiery/trainer.py:42-64)
            v = 0.1 * torch.randn((), generator=g)
        elif t.dim() == 1 and key.endswith('weight'):          # BN / affine scale
            v = 0.75 + 0.5 * torch.rand(t.shape, generator=g)
        elif t.dim() == 1:
            v = 0.1 * torch.randn(t.shape, generator=g)
        else:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.0 / fan_in ** 0.5)     # keeps activations O(1..10) through the stack
        new[key] = v.to(t.dtype)
    model.load_state_dict(new)
    return new
