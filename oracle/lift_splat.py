"""CPU restatement of the Lift-Splat frustum-to-voxel path.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Every function cites the reference lines it follows.  Integer work is numpy; the two places where
the reference's result depends on a library's floating-point behaviour are restated explicitly:

* `torch.inverse` on the CPU (LAPACK) is called as-is - torch is the library the reference calls;
* the small batched matmuls (`rotation.matmul(inverse)`, `combined.matmul(points)`) run through
  ATen's naive CPU kernel: products and sums rounded separately, k ascending, no FMA.  That order
  is written out here and checked bit-for-bit against the reference (tests/test_oracle_vs_reference.py).

PARITY PIN: the reference ships no tests or golden vectors; this oracle is
pinned against the reference's own code executed in the build container and against the fixtures
that run produced (tests/golden/, generator tests/golden/make_golden.py).
"""
import numpy as np
import torch

F32 = np.float32


def bev_parameters(x_bounds, y_bounds, z_bounds):
    """`gen_dx_bx`.  reference: fiery/utils/geometry.py:39-58.

    Returns (resolution f32[3], start_position f32[3] = first cell centre, dimension i64[3]); the
    dimension is a python-float division truncated toward zero (Y=[-9.6,9.7,0.1] gives 192).
    """
    rows = [x_bounds, y_bounds, z_bounds]
    resolution = torch.tensor([row[2] for row in rows])
    start = torch.tensor([row[0] + row[2] / 2.0 for row in rows])
    dimension = torch.tensor([(row[1] - row[0]) / row[2] for row in rows], dtype=torch.long)
    return resolution.numpy().astype(F32), start.numpy().astype(F32), dimension.numpy().astype(np.int64)


def create_frustum(final_dim, downsample, d_bound):
    """reference: fiery/models/fiery.py:109-128.  -> f32 (D, fH, fW, 3) holding (u, v, depth)."""
    h, w = final_dim
    fh, fw = h // downsample, w // downsample
    depth = torch.arange(*d_bound, dtype=torch.float)
    xs = torch.linspace(0, w - 1, fw, dtype=torch.float)
    ys = torch.linspace(0, h - 1, fh, dtype=torch.float)
    out = np.empty((depth.shape[0], fh, fw, 3), dtype=F32)
    out[..., 0] = xs.numpy()[None, None, :]
    out[..., 1] = ys.numpy()[None, :, None]
    out[..., 2] = depth.numpy()[:, None, None]
    return out


def _naive_matmul(a, b):
    """ATen CPU small-matmul order: acc = a[..,0]*b[0]; acc += a[..,k]*b[k] (each op rounded to f32)."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    k = a.shape[-1]
    acc = (a[..., :, 0, None] * b[..., None, 0, :]).astype(F32)
    for i in range(1, k):
        acc = (acc + (a[..., :, i, None] * b[..., None, i, :]).astype(F32)).astype(F32)
    return acc


def camera_matrices(intrinsics, extrinsics):
    """Per-camera `R . K^-1` and translation.  reference: fiery/models/fiery.py:195,203.

    intrinsics (..., 3, 3), extrinsics (..., 4, 4) f32 -> combined (..., 3, 3), translation (..., 3).
    """
    intrinsics = torch.as_tensor(intrinsics, dtype=torch.float32)
    extrinsics = np.asarray(extrinsics, dtype=F32)
    inverse = torch.inverse(intrinsics).numpy()
    rotation = extrinsics[..., :3, :3]
    return _naive_matmul(rotation, inverse), extrinsics[..., :3, 3].copy()


def canonical_intrinsics_inverse(intrinsics):
    """Closed form that LAPACK's result equals bit-for-bit for pinhole intrinsics with zero skew,
    K[2,2] = 1 and fx >= |cx|, fy >= |cy| (no pivoting): the form the device kernel evaluates.
    Measured on this image's torch/MKL: [0,2] = -(cx * (1/fx)) always; [1,2] = -(cy / fy) for 99.995 % of
    random calibrations (MKL's vector division is not always correctly rounded).  The rigs the tests and
    fixtures use are checked to match `torch.inverse` exactly in tests/test_oracle_cpu.py."""
    k = np.asarray(intrinsics, dtype=F32)
    inv = np.zeros_like(k)
    rfx = (F32(1.0) / k[..., 0, 0]).astype(F32)
    rfy = (F32(1.0) / k[..., 1, 1]).astype(F32)
    inv[..., 0, 0] = rfx
    inv[..., 1, 1] = rfy
    inv[..., 0, 2] = -(k[..., 0, 2] * rfx).astype(F32)
    inv[..., 1, 2] = -(k[..., 1, 2] / k[..., 1, 1]).astype(F32)     # LAPACK's path divides here, measured
    inv[..., 2, 2] = 1.0
    return inv


def get_geometry(frustum, intrinsics, extrinsics):
    """Ego-frame position of every frustum point.  reference: fiery/models/fiery.py:193-208.

    frustum (D,fH,fW,3); intrinsics (B,N,3,3); extrinsics (B,N,4,4) -> (B,N,D,fH,fW,3) f32.
    """
    frustum = np.asarray(frustum, dtype=F32)
    combined, translation = camera_matrices(intrinsics, extrinsics)
    depth = frustum[..., 2]
    p0 = (frustum[..., 0] * depth).astype(F32)      # u*d   (fiery.py:202)
    p1 = (frustum[..., 1] * depth).astype(F32)      # v*d
    p2 = depth
    out = np.empty(combined.shape[:-2] + frustum.shape, dtype=F32)
    for i in range(3):
        m = combined[..., i, :][..., None, None, None, :]       # (B,N,1,1,1,3)
        acc = (m[..., 0] * p0).astype(F32)
        acc = (acc + (m[..., 1] * p1).astype(F32)).astype(F32)
        acc = (acc + (m[..., 2] * p2).astype(F32)).astype(F32)
        out[..., i] = (acc + translation[..., i][..., None, None, None]).astype(F32)   # fiery.py:205
    return out


def voxel_indices(geometry, resolution, start, dimension):
    """Quantise, mask and rank.  reference: fiery/models/fiery.py:236-256.

    geometry (..., 3) f32 -> idx int64 (..., 3) (trunc toward zero, like `.long()`), keep bool (...),
    rank int64 (...) = ix*(Y*Z) + iy*Z + iz (meaningful where keep).
    """
    g = np.asarray(geometry, dtype=F32)
    resolution = np.asarray(resolution, dtype=F32)
    start = np.asarray(start, dtype=F32)
    nx = [int(v) for v in dimension]
    origin = (start - (resolution / F32(2.0)).astype(F32)).astype(F32)
    scaled = ((g - origin).astype(F32) / resolution).astype(F32)
    with np.errstate(invalid='ignore'):
        idx = np.trunc(scaled).astype(np.int64)          # x86 cvttss2si semantics for finite in-range values
    idx[~np.isfinite(scaled)] = np.iinfo(np.int64).min    # what .long() yields for nan/inf on x86
    keep = np.ones(idx.shape[:-1], dtype=bool)
    for axis in range(3):
        keep &= (idx[..., axis] >= 0) & (idx[..., axis] < nx[axis])
    rank = idx[..., 0] * (nx[1] * nx[2]) + idx[..., 1] * nx[2] + idx[..., 2]
    return idx, keep, rank


def voxel_pool_reference(x, geometry, resolution, start, dimension):
    """`voxel_pooling` with the prefix-sum trick, one batch element.
    reference: fiery/models/fiery.py:231-271 + fiery/utils/geometry.py:283-302 (VoxelsSumming.forward).

    x (N, C) f32 (point-major, as `x[b].reshape(N, c)`), geometry (N, 3) f32 -> (C, X, Y) f32.
    Uses torch CPU `cumsum`/`argsort` so the floating-point noise is the reference's own.
    """
    nx = [int(v) for v in dimension]
    if nx[2] != 1:
        raise ValueError('the reference only supports a single z cell (fiery/models/fiery.py:268-271)')
    idx, keep, rank = voxel_indices(geometry, resolution, start, dimension)
    xt = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)[torch.from_numpy(keep)]
    idx_k = torch.from_numpy(idx[keep])
    rank_k = torch.from_numpy(rank[keep])
    order = rank_k.argsort()
    xt, idx_k, rank_k = xt[order], idx_k[order], rank_k[order]
    csum = xt.cumsum(0)
    last = torch.ones(csum.shape[0], dtype=torch.bool)
    last[:-1] = rank_k[1:] != rank_k[:-1]
    csum, idx_k = csum[last], idx_k[last]
    sums = torch.cat((csum[:1], csum[1:] - csum[:-1]))
    bev = torch.zeros((nx[2], nx[0], nx[1], xt.shape[1]))
    bev[idx_k[:, 2], idx_k[:, 0], idx_k[:, 1]] = sums
    return bev.permute(0, 3, 1, 2).squeeze(0).numpy()


def voxel_pool_exact(x, geometry, resolution, start, dimension):
    """Same mapping, but every voxel is the float64 sum of its points: the noise-free truth the
    reference's prefix-sum only approximates ('the oracle is noisier than the truth')."""
    nx = [int(v) for v in dimension]
    idx, keep, rank = voxel_indices(geometry, resolution, start, dimension)
    x = np.asarray(x, dtype=np.float64)[keep]
    out = np.zeros((nx[0] * nx[1] * nx[2], x.shape[1]), dtype=np.float64)
    np.add.at(out, rank[keep], x)
    return out.reshape(nx[0], nx[1], nx[2], -1)[:, :, 0, :].transpose(2, 0, 1)


def lifted_to_points(lifted):
    """(n, C, D, h, w) encoder output -> (N, C) point-major matrix, N = n*D*h*w ordered (n, d, h, w):
    the `view/permute` of fiery/models/fiery.py:214-219 followed by `x[b].reshape(N, c)` (:233)."""
    lifted = np.asarray(lifted)
    n, c = lifted.shape[:2]
    return np.ascontiguousarray(np.moveaxis(lifted, 1, -1)).reshape(-1, c)


def voxel_pool_backward(grad_bev, geometry, resolution, start, dimension):
    """Gradient of `voxel_pooling` with respect to its point features, one batch element.
    reference: fiery/utils/geometry.py:304-314 (`VoxelsSumming.backward`: `grad_out[cumsum(keep) - keep]` hands every
    sorted point the gradient of its voxel), seen through what autograd does around it to undo the argsort, the bounds
    mask and the reshape of fiery/models/fiery.py:233-261, and the dense scatter + permute of :263-271.

    grad_bev (C, X, Y) f32, geometry (N, 3) f32 -> (N, C) f32: row p = grad_bev[:, ix_p, iy_p] for in-grid points, 0
    for the others.  A pure copy, so the result is bit-exact.
    """
    nx = [int(v) for v in dimension]
    if nx[2] != 1:
        raise ValueError('the reference only supports a single z cell (fiery/models/fiery.py:268-271)')
    idx, keep, _ = voxel_indices(geometry, resolution, start, dimension)
    g = np.asarray(grad_bev, dtype=F32)
    out = np.zeros((idx.shape[0], g.shape[0]), dtype=F32)
    out[keep] = g[:, idx[keep, 0], idx[keep, 1]].T
    return out


def lift_splat_backward(grad_bev, depth_prob, features, geometry, resolution, start, dimension):
    """Gradients of `voxel_pooling(depth_prob (x) features)` (fiery/models/encoder.py:99-100 followed by
    fiery/models/fiery.py:221-273) with respect to both factors, one batch element, accumulated in float64.

    depth_prob (n, D, h, w), features (n, C, h, w), geometry (n*D*h*w, 3) -> (n, D, h, w), (n, C, h, w).
    """
    n, D, h, w = depth_prob.shape
    C = features.shape[1]
    gx = voxel_pool_backward(grad_bev, geometry, resolution, start, dimension).astype(np.float64)
    gx = gx.reshape(n, D, h, w, C)
    g_depth = np.einsum('ndhwc,nchw->ndhw', gx, np.asarray(features, dtype=np.float64))
    g_feat = np.einsum('ndhwc,ndhw->nchw', gx, np.asarray(depth_prob, dtype=np.float64))
    return g_depth, g_feat
