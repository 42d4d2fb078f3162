"""The lift-splat kernel *sources* (fiery_amd/csrc/lift_splat.hip) executed on the CPU simulator and
checked against the oracle.  Covers index arithmetic, column classification, LDS tiling and both
accumulation modes without a GPU; the -m gpu tier repeats this through the real library at full size."""
import numpy as np
import pytest
import torch

from fiery_amd import native
from fiery_amd.synthetic import camera_rig
from oracle import lift_splat as ls


def _grid(xb, yb, zb):
    res, start, dim = ls.bev_parameters(xb, yb, zb)
    origin = (start - res / np.float32(2.0)).astype(np.float32)
    return native.make_grid(origin, res, dim), (res, start, dim)


def _small_problem(seed, n_cam=2, D=5, H=6, W=10, C=3, frames=2, jitter=True):
    frustum = ls.create_frustum((H * 8, W * 8), 8, [2.0, 2.0 + D * 3.0, 3.0])
    assert frustum.shape == (D, H, W, 3)
    intr, extr = camera_rig(n_cam, jitter=jitter)
    intr = intr.clone()
    intr[:, 0, 2] = W * 4.0            # principal point of the small image
    intr[:, 1, 2] = H * 4.0
    intr[:, 0, 0] = intr[:, 1, 1] = 60.0
    intr = intr.unsqueeze(0).expand(frames, -1, -1, -1).contiguous()
    extr = extr.unsqueeze(0).expand(frames, -1, -1, -1).contiguous()
    gen = torch.Generator().manual_seed(seed)
    lifted = torch.randn(frames, n_cam, C, D, H, W, generator=gen)
    return frustum, intr, extr, lifted


def test_camera_matrices_and_geometry_bit_exact(sim):
    frustum, intr, extr, _ = _small_problem(0, n_cam=6)
    cam = sim.camera_matrices(intr.reshape(-1, 3, 3).contiguous(), extr.reshape(-1, 4, 4).contiguous())
    comb, trans = ls.camera_matrices(intr.numpy(), extr.numpy())
    assert np.array_equal(cam[:, :9].numpy().reshape(-1, 3, 3), comb.reshape(-1, 3, 3))
    assert np.array_equal(cam[:, 9:].numpy(), trans.reshape(-1, 3))
    geo = sim.lift_geometry(torch.from_numpy(frustum), cam)
    want = ls.get_geometry(frustum, intr.numpy(), extr.numpy()).reshape(geo.shape)
    assert np.array_equal(geo.numpy(), want)


def test_general_intrinsics_fall_back_to_adjugate(sim):
    k = torch.tensor([[[300.0, 2.0, 310.0], [0.5, 280.0, 120.0], [0.0, 0.0, 1.0]]])
    e = torch.eye(4).unsqueeze(0)
    cam = sim.camera_matrices(k, e)
    want = torch.inverse(k)[0]
    assert torch.allclose(cam[0, :9].view(3, 3), want, rtol=1e-5, atol=1e-7)


def test_host_camera_matrices_make_geometry_bit_exact_for_any_intrinsics(sim):
    """Skewed / non-unit intrinsics (the device kernel's adjugate path is only close to LAPACK): with the 3x3 algebra
    done by `host_camera_matrices` - the reference's own CPU operators - the device product is bit-exact."""
    from fiery_amd.model import host_camera_matrices
    frustum, intr, extr, _ = _small_problem(3, n_cam=6)
    gen = torch.Generator().manual_seed(5)
    intr = intr.clone()
    intr[..., 0, 1] = 2.0 * torch.randn(intr.shape[:2], generator=gen)          # skew
    intr[..., 1, 0] = 0.3 * torch.randn(intr.shape[:2], generator=gen)
    intr[..., 2, 2] = 1.0 + 0.01 * torch.randn(intr.shape[:2], generator=gen)
    intr[..., 0, 0] *= 0.3                                                       # fx < cx: LAPACK pivots
    cam = host_camera_matrices(intr, extr)
    comb, trans = ls.camera_matrices(intr.numpy(), extr.numpy())
    assert np.array_equal(cam[:, :9].numpy().reshape(-1, 3, 3), comb.reshape(-1, 3, 3))
    geo = sim.lift_geometry(torch.from_numpy(frustum), cam)
    want = ls.get_geometry(frustum, intr.numpy(), extr.numpy()).reshape(geo.shape)
    assert np.array_equal(geo.numpy(), want)
    device_cam = sim.camera_matrices(intr.reshape(-1, 3, 3).contiguous(), extr.reshape(-1, 4, 4).contiguous())
    assert torch.allclose(device_cam, cam, rtol=1e-4, atol=1e-6)                 # the device form is close, not equal


@pytest.mark.parametrize('jitter', [True, False])
def test_voxel_index_bit_exact(sim, jitter):
    frustum, intr, extr, _ = _small_problem(1, jitter=jitter)
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 0.5], [-20.0, 20.0, 0.5], [-10.0, 10.0, 20.0])
    rank, idx = sim.voxel_index(torch.from_numpy(geo), grid)
    idx_o, keep_o, rank_o = ls.voxel_indices(geo.reshape(-1, 3), res, start, dim)
    got_rank = rank.numpy().astype(np.int64)
    assert np.array_equal(got_rank >= 0, keep_o)
    assert np.array_equal(got_rank[keep_o], rank_o[keep_o])
    assert np.array_equal(idx.numpy().astype(np.int64)[keep_o], idx_o[keep_o])
    # the reference's trunc-toward-zero band: some kept points sit in (-1, 0) cell units
    assert keep_o.any()


def test_voxel_index_special_values(sim):
    grid, (res, start, dim) = _grid([-2.0, 2.0, 0.5], [-2.0, 2.0, 0.5], [-10.0, 10.0, 20.0])
    origin = start - res / 2
    pts = np.array([
        [origin[0] - 0.25, 0.0, 0.0],      # in (-1, 0) cell units on x -> cell 0, kept
        [origin[0] - 0.5, 0.0, 0.0],       # exactly -1 cell -> dropped
        [origin[0] + 4.0, 0.0, 0.0],       # exactly nx cells -> dropped
        [np.nan, 0.0, 0.0], [np.inf, 0.0, 0.0], [-np.inf, 0.0, 0.0], [0.0, 0.0, 1e30],
        [0.0, 0.0, 0.0],
    ], dtype=np.float32)
    rank, _ = sim.voxel_index(torch.from_numpy(pts), grid)
    _, keep, rank_o = ls.voxel_indices(pts, res, start, dim)
    assert keep.tolist() == [True, False, False, False, False, False, False, True]
    got = rank.numpy()
    assert np.array_equal(got >= 0, keep)
    assert np.array_equal(got[keep], rank_o[keep])


@pytest.mark.parametrize('flags', [0, native.POOL_DETERMINISTIC])
@pytest.mark.parametrize('tile', [0, 97])
def test_voxel_pool_matches_oracle(sim, flags, tile):
    frustum, intr, extr, lifted = _small_problem(2)
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    # logical (f, n, d, h, w, c) strides of the encoder's native (f, n, C, D, H, W) layout
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid,
                         tile_voxels=tile, flags=flags)
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        ref = ls.voxel_pool_reference(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6
        assert np.abs(out[f].numpy() - ref).max() < 1e-4
        assert ((out[f].numpy() != 0) == (exact != 0)).all() or np.abs(exact[(out[f].numpy() != 0) != (exact != 0)]).max() < 1e-6


def test_voxel_pool_point_major_layout_and_mixed_columns(sim):
    """A rolled camera makes columns cross voxels (the slow, per-point branch); x given point-major."""
    frustum, intr, extr, lifted = _small_problem(3, n_cam=2, frames=1)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll        # camera 0 rotated 90 degrees about its optical axis
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 0.5], [-20.0, 20.0, 0.5], [-10.0, 10.0, 20.0])
    x_pm = lifted.permute(0, 1, 3, 4, 5, 2).contiguous()      # (f, n, d, h, w, c) physically
    out = sim.voxel_pool(x_pm, x_pm.stride(), torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    pts = ls.lifted_to_points(lifted[0].numpy())
    exact = ls.voxel_pool_exact(pts, geo[0].reshape(-1, 3), res, start, dim)
    assert np.abs(out[0].numpy() - exact).max() < 5e-6


def test_voxel_pool_empty_and_single_voxel(sim):
    frustum, intr, extr, lifted = _small_problem(4, frames=1)
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    # grid far away from every point: all zeros
    grid, _ = _grid([500.0, 510.0, 1.0], [500.0, 510.0, 1.0], [-10.0, 10.0, 20.0])
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    assert out.abs().max() == 0
    # one huge voxel swallowing everything: every channel sums all points
    grid, (res, start, dim) = _grid([-1000.0, 1000.0, 2000.0], [-1000.0, 1000.0, 2000.0], [-1000.0, 1000.0, 2000.0])
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    want = lifted[0].double().sum(dim=(0, 2, 3, 4))
    assert torch.allclose(out[0, :, 0, 0].double(), want, atol=1e-4)


def test_voxel_pool_rejects_multiple_z_cells(sim):
    frustum, intr, extr, lifted = _small_problem(5, frames=1)
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, _ = _grid([-20.0, 20.0, 1.0], [-20.0, 20.0, 1.0], [-10.0, 10.0, 10.0])
    st = lifted.stride()
    with pytest.raises(native.NativeError, match='one z cell'):
        sim.voxel_pool(lifted, (st[0], st[1], st[3], st[4], st[5], st[2]), torch.from_numpy(geo),
                       frames, n_cam, D, H, W, C, grid)


def test_fused_lift_splat_matches_unfused(sim):
    frustum, intr, extr, _ = _small_problem(6)
    frames, n_cam, D, H, W, C = 2, 2, 5, 6, 10, 4
    gen = torch.Generator().manual_seed(7)
    logits = torch.randn(frames * n_cam, D, H, W, generator=gen)
    feats = torch.randn(frames * n_cam, C, H, W, generator=gen)
    prob = sim.depth_softmax(logits)
    assert torch.allclose(prob, logits.softmax(dim=1), atol=1e-6)
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    out = sim.lift_splat(prob, feats, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    lifted = (prob.unsqueeze(1) * feats.unsqueeze(2)).view(frames, n_cam, C, D, H, W)
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6


@pytest.mark.parametrize('flags', [0, native.POOL_DETERMINISTIC])
@pytest.mark.parametrize('rolled', [False, True])
def test_quad_path_16_byte_rows(sim, flags, rolled):
    """W divisible by 4 with the encoder's native layout takes the vectorised kernel (four columns per
    work-item); `rolled` makes camera 0's columns cross voxels, which drives its run-length branch."""
    frustum, intr, extr, lifted = _small_problem(8, n_cam=2, W=12, C=3, frames=2)
    if rolled:
        roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
        extr = extr.clone()
        extr[:, 0] = extr[:, 0] @ roll
    frames, n_cam, C, D, H, W = lifted.shape
    assert W % 4 == 0
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 0.5], [-16.0, 16.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    out = sim.voxel_pool(lifted, (st[0], st[1], st[3], st[4], st[5], st[2]), torch.from_numpy(geo), frames, n_cam, D, H, W, C,
                         grid, tile_voxels=301, flags=flags)
    for f in range(frames):
        exact = ls.voxel_pool_exact(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6


def test_many_units_and_general_items(sim):
    """3 frames x 6 channels x 2 tiles on the 1-D unit grid, with camera 0 rolled so that some quads hold a
    column of four or more runs (their own list, row-by-row path)."""
    frustum, intr, extr, lifted = _small_problem(21, n_cam=2, W=12, C=6, frames=3)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 0.5], [-16.0, 16.0, 0.5], [-10.0, 10.0, 20.0])
    q = np.floor((geo[0, 0] - (start - res / np.float32(2))) / res).astype(np.int64)       # camera 0, frame 0
    r = q[..., 0] * int(dim[1]) + q[..., 1]
    runs = (r[:, 1:, :] != r[:, :-1, :]).sum(axis=1) + 1
    assert runs.max() >= 4                                   # the general path is exercised
    st = lifted.stride()
    out = sim.voxel_pool(lifted, (st[0], st[1], st[3], st[4], st[5], st[2]), torch.from_numpy(geo), frames, n_cam, D, H, W, C,
                         grid, tile_voxels=2560)
    for f in range(frames):
        exact = ls.voxel_pool_exact(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6


def test_quad_path_fused(sim):
    frustum, intr, extr, _ = _small_problem(9, W=12)
    frames, n_cam, D, H, W, C = 2, 2, 5, 6, 12, 4
    gen = torch.Generator().manual_seed(10)
    prob = torch.randn(frames * n_cam, D, H, W, generator=gen).softmax(dim=1)
    feats = torch.randn(frames * n_cam, C, H, W, generator=gen)
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    out = sim.lift_splat(prob, feats, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    lifted = (prob.unsqueeze(1) * feats.unsqueeze(2)).view(frames, n_cam, C, D, H, W)
    for f in range(frames):
        exact = ls.voxel_pool_exact(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6


# ------------------------------------------------------------------------------------------------------
# backward (training): fiery_voxel_pool_bwd / fiery_lift_splat_bwd / fiery_depth_softmax_bwd
# ------------------------------------------------------------------------------------------------------
def _bwd_case(seed, W, rolled=False, **kw):
    frustum, intr, extr, lifted = _small_problem(seed, W=W, **kw)
    if rolled:
        roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
        extr = extr.clone()
        extr[:, 0] = extr[:, 0] @ roll
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    frames, n_cam, C, D, H, W = lifted.shape
    g = torch.randn(frames, C, int(dim[0]), int(dim[1]), generator=torch.Generator().manual_seed(seed + 100))
    return lifted, geo, grid, (res, start, dim), g


@pytest.mark.parametrize('W,layout', [(10, 'native'), (12, 'native'), (12, 'point_major')])
def test_voxel_pool_bwd_is_a_bit_exact_gather(sim, W, layout):
    """Both the scalar (W = 10) and the 16-byte (W = 12) form, into the encoder's layout and into a point-major one,
    with the ranks the forward call left in its workspace."""
    lifted, geo, grid, (res, start, dim), g = _bwd_case(31, W, rolled=True)
    frames, n_cam, C, D, H, W = lifted.shape
    st = lifted.stride()
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    sim.voxel_pool(lifted, (st[0], st[1], st[3], st[4], st[5], st[2]), torch.from_numpy(geo), frames, n_cam, D, H, W, C,
                   grid, workspace=ws)
    rank = ws[:frames * n_cam * D * H * W]
    if layout == 'native':
        gx = torch.full((frames, n_cam, C, D, H, W), float('nan')).permute(0, 1, 3, 4, 5, 2)
    else:
        gx = torch.full((frames, n_cam, D, H, W, C), float('nan'))
    sim.voxel_pool_bwd(g, rank, frames, n_cam, D, H, W, C, gx)
    for f in range(frames):
        want = ls.voxel_pool_backward(g[f].numpy(), geo[f].reshape(-1, 3), res, start, dim)
        assert np.array_equal(gx[f].reshape(-1, C).numpy(), want)
    assert (gx != 0).any() and (gx == 0).any()


def test_voxel_pool_bwd_with_ranks_from_voxel_index(sim):
    lifted, geo, grid, (res, start, dim), g = _bwd_case(32, 12, frames=1)
    frames, n_cam, C, D, H, W = lifted.shape
    rank, _ = sim.voxel_index(torch.from_numpy(geo), grid, want_idx=False)
    gx = torch.empty(frames, n_cam, C, D, H, W).permute(0, 1, 3, 4, 5, 2)
    sim.voxel_pool_bwd(g, rank, frames, n_cam, D, H, W, C, gx)
    want = ls.voxel_pool_backward(g[0].numpy(), geo[0].reshape(-1, 3), res, start, dim)
    assert np.array_equal(gx[0].reshape(-1, C).numpy(), want)


def test_lift_splat_bwd_matches_oracle(sim):
    frustum, intr, extr, _ = _small_problem(33, W=12, C=5)
    frames, n_cam, D, H, W, C = 2, 2, 5, 6, 12, 5
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    gen = torch.Generator().manual_seed(34)
    logits = torch.randn(frames * n_cam, D, H, W, generator=gen)
    feats = torch.randn(frames, n_cam, C, H, W, generator=gen)
    g = torch.randn(frames, C, int(dim[0]), int(dim[1]), generator=gen)
    prob = sim.depth_softmax(logits)
    rank, _ = sim.voxel_index(torch.from_numpy(geo), grid, want_idx=False)
    gd, gf = sim.lift_splat_bwd(g, rank, prob, feats, frames, n_cam, D, H, W, C)
    gl = sim.depth_softmax_bwd(prob, gd)
    p64 = prob.double().view(frames, n_cam, D, H, W).numpy()
    for f in range(frames):
        wd, wf = ls.lift_splat_backward(g[f].numpy(), p64[f], feats[f].numpy(), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(gd.view(frames, n_cam, D, H, W)[f].numpy() - wd).max() < 1e-5
        assert np.abs(gf[f].numpy() - wf).max() < 1e-5
        wl = p64[f] * (wd - (p64[f] * wd).sum(axis=1, keepdims=True))
        assert np.abs(gl.view(frames, n_cam, D, H, W)[f].numpy() - wl).max() < 1e-5
    only_feat = sim.lift_splat_bwd(g, rank, prob, feats, frames, n_cam, D, H, W, C, want_depth=False)
    assert only_feat[0] is None and torch.equal(only_feat[1], gf)


def test_backward_against_the_reference_fixture(sim):
    """tests/golden/pooling_bwd_small.npz: the reference's own autograd (generator: tests/golden/make_golden.py)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pooling_bwd_small.npz'))
    from tests.helpers import tiny_cfg
    cfg = tiny_cfg('baseline.yml', bev=16)
    grid, _ = _grid(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
    geo = torch.from_numpy(gold['geometry'])
    frames, n_cam, D, H, W = geo.shape[:5]
    C = gold['features'].shape[1]
    g = torch.from_numpy(gold['grad_bev'])
    rank, _ = sim.voxel_index(geo, grid, want_idx=False)
    gx = torch.empty(frames, n_cam, C, D, H, W).permute(0, 1, 3, 4, 5, 2)
    sim.voxel_pool_bwd(g, rank, frames, n_cam, D, H, W, C, gx)
    want = torch.from_numpy(gold['grad_lifted']).view(frames, n_cam, C, D, H, W).permute(0, 1, 3, 4, 5, 2)
    assert torch.equal(gx, want)
    prob = sim.depth_softmax(torch.from_numpy(gold['depth_logits']))
    feats = torch.from_numpy(gold['features']).view(frames, n_cam, C, H, W)
    gd, gf = sim.lift_splat_bwd(g, rank, prob, feats, frames, n_cam, D, H, W, C)
    gl = sim.depth_softmax_bwd(prob, gd)
    assert np.abs(gl.numpy() - gold['grad_depth_logits']).max() < 1e-5
    assert np.abs(gf.reshape(-1, C, H, W).numpy() - gold['grad_features']).max() < 1e-5


def test_backward_rejects_bad_arguments(sim):
    g = torch.zeros(1, 2, 4, 4)
    rank = torch.zeros(8, dtype=torch.int32)
    with pytest.raises(native.NativeError):
        sim.voxel_pool_bwd(g, rank, 1, 1, 1, 2, 4, 0, torch.empty(1, 1, 1, 2, 4, 1))


@pytest.mark.parametrize('H,batch', [(14, 7), (28, 7), (28, 14)])
@pytest.mark.parametrize('flags', [0, native.POOL_DETERMINISTIC])
def test_pipelined_walk_has_the_bits_of_the_plain_loop(sim, monkeypatch, H, batch, flags):
    """Columns whose height is an even number of batches take the software-pipelined walk (two alternating row
    buffers, packed run sums).  A slightly pitched and a rolled camera give two-, three- and many-run columns; enough
    items per thread that the cross-item prefetch runs.  Against the oracle, and bit for bit against the plain loop
    (in the order-independent fixed-point mode; the fp32 mode adds the same run sums in a thread order of its own)."""
    frustum, intr, extr, lifted = _small_problem(40, n_cam=3, D=16, H=H, W=40, C=2, frames=2)      # 480 items, 256 threads
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    a = 0.06
    pitch = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, float(np.cos(a)), -float(np.sin(a)), 0.0],
                          [0.0, float(np.sin(a)), float(np.cos(a)), 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    extr[:, 1] = extr[:, 1] @ pitch
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-30.0, 30.0, 0.5], [-24.0, 24.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    monkeypatch.setenv('FIERY_POOL_BATCH', str(batch))
    monkeypatch.setenv('FIERY_POOL_PLANE', '0')                  # the tiled kernel: the whole-plane form has no pipelined walk
    monkeypatch.setenv('FIERY_POOL_COMPACT', '0')                # ... and neither has the compact-plane form
    monkeypatch.setenv('FIERY_POOL_PIPE', '1')
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, flags=flags)
    monkeypatch.setenv('FIERY_POOL_PIPE', '0')
    plain = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, flags=flags)
    if flags:
        assert torch.equal(out, plain)
    else:
        assert (out - plain).abs().max() < 2e-6
    for f in range(frames):
        exact = ls.voxel_pool_exact(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 5e-6
    assert (out != 0).any()


@pytest.mark.parametrize('H,batch,queue_cap,tail_parts', [(28, 7, None, 0), (28, 7, 3, 3), (12, 16, None, 2), (12, 8, 0, 0)])
def test_whole_plane_form(sim, monkeypatch, H, batch, queue_cap, tail_parts):
    """The list-free kernel with compact 8-byte descriptors (one tile, fp32 cells, 16-byte rows): a rolled camera
    fills the queue of many-run quads (a tiny queue makes the rest walk their rows on the spot), a pitched one gives
    two- and three-run columns, part of the frustum lies outside the grid; `tail_parts` cuts the last units into
    several workgroups that add their partial planes to the output.  Against the oracle and against the tiled kernel
    on the same inputs."""
    frustum, intr, extr, lifted = _small_problem(50, n_cam=3, D=16, H=H, W=40, C=2, frames=2)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    a = 0.06
    pitch = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, float(np.cos(a)), -float(np.sin(a)), 0.0],
                          [0.0, float(np.sin(a)), float(np.cos(a)), 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    extr[:, 1] = extr[:, 1] @ pitch
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    monkeypatch.setenv('FIERY_POOL_BATCH', str(batch))
    if queue_cap is not None:
        monkeypatch.setenv('FIERY_POOL_QUEUE_CAP', str(queue_cap))
    if tail_parts:
        monkeypatch.setenv('FIERY_POOL_TAIL_PARTS', str(tail_parts))
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    garbage = torch.full((frames, C, int(dim[0]), int(dim[1])), 7.0)
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, out=garbage, workspace=ws)
    rank_left = ws[:frames * n_cam * D * H * W].clone()
    monkeypatch.setenv('FIERY_POOL_PLANE', '0')
    tiled = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    assert (out - tiled).abs().max() < 2e-5          # fp32 LDS atomics arrive in a different order in the two kernels
    kept = 0
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 2e-5          # hundreds of near-field points per voxel here
        assert np.abs(tiled[f].numpy() - exact).max() < 2e-5
        _, keep, rank_o = ls.voxel_indices(geo[f].reshape(-1, 3), res, start, dim)
        got = rank_left.view(frames, -1)[f].numpy().astype(np.int64)
        assert np.array_equal(got >= 0, keep) and np.array_equal(got[keep], rank_o[keep])      # the ranks backward needs
        kept += keep.mean() / frames
    assert 0.2 < kept < 0.95                      # some quads are skipped without being read, some are not


@pytest.mark.parametrize('H,cells,tail_parts,big_grid', [(28, 0, 0, False), (28, 576, 0, False), (28, 576, 3, False),
                                                         (14, 0, 2, False), (27, 600, 0, False), (28, 0, 0, True)])
def test_compact_plane_form(sim, monkeypatch, H, cells, tail_parts, big_grid):
    """The compact-plane kernel (LDS cells for occupied voxels only, rows dealt to four lane groups): a rolled camera
    gives many-run quads (the per-element path), a pitched one two- and three-run columns, part of the frustum lies
    outside the grid.  `cells` below the number of occupied voxels forces several passes over the rows; `tail_parts`
    cuts the last units into workgroups that add to the output; the big grid (more than 65,535 voxels) takes the
    16-byte descriptors; H = 27 / 14 leaves some lanes' last rows past the end of the column.  Against the float64
    pooling, the dense whole-plane kernel, and the oracle's ranks; the occupied-voxel counts the kernel leaves in the
    workspace must equal the oracle's."""
    frustum, intr, extr, lifted = _small_problem(60, n_cam=3, D=16, H=H, W=40, C=2, frames=2)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    a = 0.06
    pitch = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, float(np.cos(a)), -float(np.sin(a)), 0.0],
                          [0.0, float(np.sin(a)), float(np.cos(a)), 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    extr[:, 1] = extr[:, 1] @ pitch
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    if big_grid:
        grid, (res, start, dim) = _grid([-14.0, 30.0, 0.125], [-24.0, 10.0, 0.125], [-10.0, 10.0, 20.0])     # 352 x 272
    else:
        grid, (res, start, dim) = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    if cells:
        monkeypatch.setenv('FIERY_POOL_CELLS', str(cells))
    if tail_parts:
        monkeypatch.setenv('FIERY_POOL_TAIL_PARTS', str(tail_parts))
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    ws.fill_(-7)                                                           # garbage: the call must not rely on a clean workspace
    garbage = torch.full((frames, C, int(dim[0]), int(dim[1])), 7.0)
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, out=garbage, workspace=ws)
    occupied = sim.pool_occupied(ws, frames, n_cam, D, H, W, grid).clone()
    rank_left = ws[:frames * n_cam * D * H * W].clone()
    monkeypatch.setenv('FIERY_POOL_COMPACT', '0')
    dense = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    assert (out - dense).abs().max() < 2e-5          # fp32 LDS atomics arrive in a different order in the two kernels
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 2e-5
        _, keep, rank_o = ls.voxel_indices(geo[f].reshape(-1, 3), res, start, dim)
        got = rank_left.view(frames, -1)[f].numpy().astype(np.int64)
        assert np.array_equal(got >= 0, keep) and np.array_equal(got[keep], rank_o[keep])
        assert int(occupied[f]) == len(np.unique(rank_o[keep]))              # this IS the compact form, and it counted right
        if cells:
            assert int(occupied[f]) > cells                                    # several passes were needed
    assert (out != 0).any()


@pytest.mark.parametrize('case', range(24))
def test_pooling_random_shapes_against_the_oracle(sim, case):
    """Seeded sweep over camera counts, depth bins, feature-map sizes (row counts that do and do not divide the row
    batches, quads and scalar columns), channel counts, frame counts, grid sizes (one tile, several tiles) and camera
    attitudes: whichever kernel form the library picks, the sums equal the float64 pooling and the ranks left in the
    workspace equal the oracle's; the backward gather from those ranks is checked on the way."""
    rng = np.random.RandomState(1000 + case)
    n_cam, D = int(rng.randint(1, 4)), int(rng.randint(1, 10))
    H = int(rng.choice([1, 5, 7, 14, 16, 28]))
    W = int(rng.choice([4, 8, 10, 12, 20]))
    C, frames = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    frustum, intr, extr, lifted = _small_problem(2000 + case, n_cam=n_cam, D=D, H=H, W=W, C=C, frames=frames,
                                                 jitter=bool(rng.randint(2)))
    if rng.randint(3) == 0:                                            # roll one camera: many-run columns
        roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
        extr = extr.clone()
        extr[:, 0] = extr[:, 0] @ roll
    half = float(rng.choice([6.0, 15.0, 40.0]))
    step = float(rng.choice([0.5, 1.0, 2.0]))
    grid, (res, start, dim) = _grid([-half, half, step], [-half * 0.75, half * 0.75, step], [-10.0, 10.0, 20.0])
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    tile = int(rng.choice([0, 0, 61]))                                 # default plan, or a small tile: several tiles
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid, tile)
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, workspace=ws,
                         tile_voxels=tile)
    rank = ws[:frames * n_cam * D * H * W]
    g = torch.randn(frames, C, int(dim[0]), int(dim[1]), generator=torch.Generator().manual_seed(case))
    gx = torch.full((frames, n_cam, C, D, H, W), float('nan')).permute(0, 1, 3, 4, 5, 2)
    sim.voxel_pool_bwd(g, rank, frames, n_cam, D, H, W, C, gx)
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 2e-5, (case, n_cam, D, H, W, C, frames, tile)
        _, keep, rank_o = ls.voxel_indices(geo[f].reshape(-1, 3), res, start, dim)
        got = rank.view(frames, -1)[f].numpy().astype(np.int64)
        assert np.array_equal(got >= 0, keep) and np.array_equal(got[keep], rank_o[keep])
        want_gx = ls.voxel_pool_backward(g[f].numpy(), geo[f].reshape(-1, 3), res, start, dim)
        assert np.array_equal(gx[f].reshape(-1, C).numpy(), want_gx)


@pytest.mark.parametrize('case', range(12))
def test_fused_and_reproducible_forms_random_shapes(sim, case):
    """The same sweep for the fused lift (x) splat (depth softmax times features, never materialised) and for the
    order-independent fixed-point mode, which must also give identical bits on a second call."""
    rng = np.random.RandomState(3000 + case)
    n_cam, D = int(rng.randint(1, 4)), int(rng.randint(2, 9))
    H = int(rng.choice([4, 7, 8, 14, 28]))
    W = int(rng.choice([4, 8, 10, 12]))
    C, frames = int(rng.randint(1, 4)), int(rng.randint(1, 3))
    frustum, intr, extr, _ = _small_problem(4000 + case, n_cam=n_cam, D=D, H=H, W=W, C=C, frames=frames)
    grid, (res, start, dim) = _grid([-20.0, 20.0, 1.0], [-16.0, 16.0, 1.0], [-10.0, 10.0, 20.0])
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    gen = torch.Generator().manual_seed(case)
    logits = torch.randn(frames * n_cam, D, H, W, generator=gen)
    feats = torch.randn(frames, n_cam, C, H, W, generator=gen)
    prob = sim.depth_softmax(logits)
    lifted = prob.view(frames, n_cam, 1, D, H, W) * feats.view(frames, n_cam, C, 1, H, W)
    flags = native.POOL_DETERMINISTIC if case % 2 else 0
    fused = sim.lift_splat(prob, feats, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, flags=flags)
    st = lifted.stride()
    plain = sim.voxel_pool(lifted, (st[0], st[1], st[3], st[4], st[5], st[2]), torch.from_numpy(geo), frames, n_cam, D, H, W,
                           C, grid, flags=flags)
    for f in range(frames):
        exact = ls.voxel_pool_exact(ls.lifted_to_points(lifted[f].numpy()), geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(fused[f].numpy() - exact).max() < 5e-6 and np.abs(plain[f].numpy() - exact).max() < 5e-6
    if flags:
        again = sim.lift_splat(prob, feats, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, flags=flags)
        assert torch.equal(again, fused)


@pytest.mark.parametrize('persistent,C,frames', [(1, 6, 2), (0, 6, 2), (1, 5, 3), (1, 9, 1)])
def test_compact_plane_form_persistent_workgroups(sim, monkeypatch, persistent, C, frames):
    """More (channel, frame) units than the (simulated 4-CU) device has workgroup slots: the default launch is one workgroup
    per slot, each taking whole units and then a part of a tail unit in turn - a workgroup's items lie in different frames
    (the bit map is rebuilt) or in the same one (it is kept).  Same results as one workgroup per item, as the dense form and
    as the float64 pooling."""
    frustum, intr, extr, lifted = _small_problem(61, n_cam=2, D=12, H=28, W=40, C=C, frames=frames)
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    monkeypatch.setenv('FIERY_POOL_PERSISTENT', str(persistent))
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    ws.fill_(-3)
    garbage = torch.full((frames, C, int(dim[0]), int(dim[1])), 7.0)
    out = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, out=garbage, workspace=ws)
    occupied = sim.pool_occupied(ws, frames, n_cam, D, H, W, grid).clone()
    monkeypatch.setenv('FIERY_POOL_COMPACT', '0')
    dense = sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid)
    assert (out - dense).abs().max() < 2e-5
    for f in range(frames):
        pts = ls.lifted_to_points(lifted[f].numpy())
        exact = ls.voxel_pool_exact(pts, geo[f].reshape(-1, 3), res, start, dim)
        assert np.abs(out[f].numpy() - exact).max() < 2e-5
        _, keep, rank_o = ls.voxel_indices(geo[f].reshape(-1, 3), res, start, dim)
        assert int(occupied[f]) == len(np.unique(rank_o[keep]))


@pytest.mark.parametrize('H,form', [(28, 'compact'), (27, 'compact'), (14, 'compact'), (28, 'plane'), (28, 'tiled'), (31, 'tiled')])
def test_four_lane_prepass_writes_what_the_one_lane_prepass_writes(sim, monkeypatch, H, form):
    """The prepass with four lanes per column (the default when a column has at most 32 rows) against the one-thread-per-column
    form: ranks, column descriptors / quad records, tile masks, occupancy bytes and live masks - the workspace byte for byte
    up to the counters - and the pooled output, for each descriptor form, with rolled and pitched cameras (many-run and
    two- / three-run columns) and rows past the end of a part (H = 27, 14, 31)."""
    frustum, intr, extr, lifted = _small_problem(62, n_cam=3, D=10, H=H, W=40, C=2, frames=2)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    a = 0.06
    pitch = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, float(np.cos(a)), -float(np.sin(a)), 0.0],
                          [0.0, float(np.sin(a)), float(np.cos(a)), 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    extr[:, 1] = extr[:, 1] @ pitch
    frames, n_cam, C, D, H, W = lifted.shape
    geo = ls.get_geometry(frustum, intr.numpy(), extr.numpy())
    grid, (res, start, dim) = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    if form != 'compact':
        monkeypatch.setenv('FIERY_POOL_COMPACT', '0')
    if form == 'tiled':
        monkeypatch.setenv('FIERY_POOL_PLANE', '0')
    outs, spaces = [], []
    for lanes in ('1', '4'):
        monkeypatch.setenv('FIERY_POOL_PREPASS_LANES', lanes)
        ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
        ws.zero_()                                                       # (regions a form does not write must compare equal)
        outs.append(sim.voxel_pool(lifted, strides, torch.from_numpy(geo), frames, n_cam, D, H, W, C, grid, workspace=ws))
        end = sim.dll.fiery_voxel_pool_occupied_offset(frames, n_cam, D, H, W, grid.dim[0] * grid.dim[1], 0, 0) // 4 - 64
        spaces.append(ws[:end].clone())
    assert torch.equal(spaces[0], spaces[1])
    assert torch.equal(outs[0], outs[1])
    for f in range(frames):
        _, keep, rank_o = ls.voxel_indices(geo[f].reshape(-1, 3), res, start, dim)
        got = spaces[1][:frames * n_cam * D * H * W].view(frames, -1)[f].numpy().astype(np.int64)
        assert np.array_equal(got >= 0, keep) and np.array_equal(got[keep], rank_o[keep])


@pytest.mark.parametrize('persistent,channels', [(0, 5), (1, 5), (0, 19)])
def test_pooling_leaves_its_workspace_clean(sim, monkeypatch, persistent, channels):
    """FIERY_POOL_WORKSPACE_CLEAN: on a zero-filled workspace the compact form skips its memset; its last workgroup re-zeroes
    the occupancy bytes, live masks and counters, so the NEXT call (another rig: other voxels occupied, other quads live) may
    skip it too.  Results equal the unflagged calls' on a garbage workspace.  (10 and 38 (channel, frame) units: fewer than the
    sixteen completion counters, and more with a ragged last round.)"""
    from fiery_amd import native
    monkeypatch.setenv('FIERY_POOL_PERSISTENT', str(persistent))
    grid, (res, start, dim) = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    ws = None
    for seed in (63, 64, 65):
        frustum, intr, extr, lifted = _small_problem(seed, n_cam=2, D=12, H=28, W=40, C=channels, frames=2)
        frames, n_cam, C, D, H, W = lifted.shape
        geo = torch.from_numpy(ls.get_geometry(frustum, intr.numpy(), extr.numpy()))
        st = lifted.stride()
        strides = (st[0], st[1], st[3], st[4], st[5], st[2])
        if ws is None:
            ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid, zeroed=True)
        got = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=ws, flags=native.POOL_WORKSPACE_CLEAN)
        dirty = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
        dirty.fill_(-5)
        want = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=dirty)
        assert torch.equal(got, want)
        off = sim.dll.fiery_voxel_pool_occupied_offset(frames, n_cam, D, H, W, grid.dim[0] * grid.dim[1], 0, 0) // 4
        assert torch.equal(sim.pool_occupied(ws, frames, n_cam, D, H, W, grid), sim.pool_occupied(dirty, frames, n_cam, D, H, W, grid))
        n_occ_words = frames * ((grid.dim[0] * grid.dim[1] + 63) // 64) * 16 + frames * n_cam * D + 64
        assert int(ws[off - n_occ_words:off].abs().sum()) == 0           # occupancy bytes, live masks, counters: all zero again


@pytest.mark.parametrize('H', [28, 27])
def test_no_ranks_flag_leaves_the_result_unchanged(sim, H):
    """FIERY_POOL_NO_RANKS: the prepass writes the voxel ranks of the many-run quads only (the compact kernel walks those
    row by row); planes are bit-identical to the call that leaves all ranks behind, on a rig with a rolled camera (many-run
    quads) and points outside the grid, on a workspace whose rank area holds garbage."""
    from fiery_amd import native
    frustum, intr, extr, lifted = _small_problem(81, n_cam=3, D=16, H=H, W=40, C=2, frames=2)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    frames, n_cam, C, D, H, W = lifted.shape
    geo = torch.from_numpy(ls.get_geometry(frustum, intr.numpy(), extr.numpy()))
    grid, _ = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    ws.fill_(-11)
    want = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=ws)
    n_pts = frames * n_cam * D * H * W
    ranks = ws[:n_pts].clone()
    ws2 = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    ws2.fill_(123456)                                             # a rank that would index far outside the bit map
    got = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=ws2, flags=native.POOL_NO_RANKS)
    assert torch.equal(got, want)
    written = ws2[:n_pts] != 123456
    assert 0 < int(written.sum()) < n_pts // 4                    # some quads are many-run quads, most are not
    assert torch.equal(ws2[:n_pts][written], ranks[written])


@pytest.mark.parametrize('records', ['narrow', 'wide'])
@pytest.mark.parametrize('flags', [0, 4])
def test_lean_prepass_writes_what_the_general_prepass_writes(sim, monkeypatch, flags, records):
    """`k_rank_columns4_lean` (H = 28, whole workgroups of columns, power-of-two cells, one z cell: the shipped configurations)
    against `k_rank_columns4` on the same rig - a rolled camera for many-run quads, a pitched one for two- and three-run
    columns, points outside the grid: the whole workspace (voxel ranks - all of them, or the many-run quads' under
    FIERY_POOL_NO_RANKS = 4 -, quad records, what the launch leaves of occupancy / live masks / counters, occupied counts)
    and the planes, byte for byte."""
    frustum, intr, extr, lifted = _small_problem(91, n_cam=3, D=16, H=28, W=40, C=2, frames=2)
    roll = torch.tensor([[0.0, -1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    a = 0.06
    pitch = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, float(np.cos(a)), -float(np.sin(a)), 0.0],
                          [0.0, float(np.sin(a)), float(np.cos(a)), 0.0], [0.0, 0.0, 0.0, 1.0]])
    extr = extr.clone()
    extr[:, 0] = extr[:, 0] @ roll
    extr[:, 1] = extr[:, 1] @ pitch
    frames, n_cam, C, D, H, W = lifted.shape
    geo = torch.from_numpy(ls.get_geometry(frustum, intr.numpy(), extr.numpy()))
    # ('wide': a grid of 65,536 voxels - 32-bit ranks in 64-byte quad records, pon_setting.yml's case)
    grid, _ = (_grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0]) if records == 'narrow' else
               _grid([-34.0, 30.0, 0.125], [-24.0, 8.0, 0.25], [-10.0, 10.0, 20.0]))
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    got = {}
    for lean in ('0', '1'):
        monkeypatch.setenv('FIERY_POOL_PREPASS_LEAN', lean)
        ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
        ws.fill_(-13)
        out = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=ws, flags=flags)
        got[lean] = (ws.clone(), out.clone())
    assert torch.equal(got['0'][0], got['1'][0])
    assert torch.equal(got['0'][1], got['1'][1])
    n_pts = frames * n_cam * D * H * W
    written = got['1'][0][:n_pts] != -13
    assert (written.all() if flags == 0 else 0 < int(written.sum()) < n_pts // 4)


@pytest.mark.parametrize('tail_parts', [0, 3])
def test_workspace_cleaning_frame_by_frame_with_more_than_sixteen_frames(sim, monkeypatch, tail_parts):
    """The compact form cleans its workspace frame by frame: the last item of a frame CLASS (frames f, f + 16, ... share one of
    sixteen ticket counters) zeroes those frames' occupancy bytes and live masks.  Eighteen frames put two frames on classes 0
    and 1; with `tail_parts` the last units are cut into parts whose tickets count too.  Two calls in a row on one zero-filled
    workspace under FIERY_POOL_WORKSPACE_CLEAN: same planes as the unflagged call on garbage, region all zero after each."""
    from fiery_amd import native
    if tail_parts:
        monkeypatch.setenv('FIERY_POOL_TAIL_PARTS', str(tail_parts))
    grid, _ = _grid([-14.0, 30.0, 0.5], [-24.0, 10.0, 0.5], [-10.0, 10.0, 20.0])
    frustum, intr, extr, lifted = _small_problem(97, n_cam=1, D=6, H=28, W=16, C=3, frames=18)
    extr = extr.clone()
    extr[:, :, 0, 3] += torch.linspace(-3.0, 3.0, 18).view(18, 1)          # every frame occupies other voxels
    frames, n_cam, C, D, H, W = lifted.shape
    geo = torch.from_numpy(ls.get_geometry(frustum, intr.numpy(), extr.numpy()))
    st = lifted.stride()
    strides = (st[0], st[1], st[3], st[4], st[5], st[2])
    dirty = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid)
    dirty.fill_(-5)
    want = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=dirty)
    ws = sim.pool_workspace(frames, n_cam, D, H, W, lifted.device, grid, zeroed=True)
    off = sim.dll.fiery_voxel_pool_occupied_offset(frames, n_cam, D, H, W, grid.dim[0] * grid.dim[1], 0, 0) // 4
    n_region = frames * ((grid.dim[0] * grid.dim[1] + 63) // 64) * 16 + frames * n_cam * D + 64
    for _ in range(2):
        got = sim.voxel_pool(lifted, strides, geo, frames, n_cam, D, H, W, C, grid, workspace=ws, flags=native.POOL_WORKSPACE_CLEAN)
        assert (got - want).abs().max() < 2e-5 and (tail_parts or torch.equal(got, want))
        assert int(ws[off - n_region:off].abs().sum()) == 0
