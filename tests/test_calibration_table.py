"""Calibration table (fiery_amd/calibration.py + `fiery_camera_matrices_cached`): the reference's LAPACK camera matrices served
to a capturable device lookup.  CPU tier: the kernel source on the simulator; GPU tier: the same through the real library and
inside a captured hipGraph, on intrinsics the device closed form does not cover (reference: fiery/models/fiery.py:193-208)."""
import numpy as np
import pytest
import torch

from fiery_amd import calibration, native
from fiery_amd.calibration import CalibrationTable
from fiery_amd.model import Fiery, host_camera_matrices
from fiery_amd.synthetic import camera_rig, make_inputs
from oracle import lift_splat as ls


def _skewed(n, seed):
    """n calibrations outside the closed form: skew, K[1,0] != 0, K[2,2] != 1, fx < cx (LAPACK pivots)."""
    gen = torch.Generator().manual_seed(seed)
    K, E = camera_rig(6, jitter=True)
    K = K[torch.arange(n) % 6].clone()
    E = E[torch.arange(n) % 6].clone()
    K[:, 0, 1] = 2.0 * torch.randn(n, generator=gen)
    K[:, 1, 0] = 0.3 * torch.randn(n, generator=gen)
    K[:, 2, 2] = 1.0 + 0.01 * torch.randn(n, generator=gen)
    K[:, 0, 0] *= 0.3
    E[:, :3, 3] += torch.randn(n, 3, generator=gen)
    return K.contiguous(), E.contiguous()


def _oracle(K, E):
    comb, trans = ls.camera_matrices(K.numpy(), E.numpy())
    return np.concatenate([comb.reshape(-1, 9), trans.reshape(-1, 3)], axis=1)


def test_host_matrices_do_not_depend_on_the_batch_they_are_computed_in():
    """An entry is made from whatever batch its calibration arrives in (a rig at set-up, one camera from a miss list): the
    reference's operators must give the same twelve numbers either way."""
    K, E = _skewed(12, 1)
    whole = host_camera_matrices(K, E)
    for i in range(12):
        assert torch.equal(host_camera_matrices(K[i:i + 1], E[i:i + 1])[0], whole[i])
    assert np.array_equal(whole.numpy(), _oracle(K, E))


def test_header_constants_match_the_host_side():
    import os
    import re
    text = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'fiery_hip.h')).read()
    for name, value in (('KEY_WORDS', calibration.KEY_WORDS), ('ENTRY_WORDS', calibration.ENTRY_WORDS),
                        ('MISS_WORDS', calibration.MISS_WORDS), ('MISS_HEADER', calibration.MISS_HEADER),
                        ('PROBES', calibration.PROBES)):
        assert int(re.search(r'#define FIERY_CALIB_%s (\d+)' % name, text).group(1)) == value


def test_primed_calibrations_are_served_bit_exactly(sim):
    K, E = _skewed(54, 2)
    table = CalibrationTable(sim, 'cpu')
    assert table.prime(K, E) == 54 and table.prime(K, E) == 0
    cam = table.lookup(K, E)
    assert np.array_equal(cam.numpy(), _oracle(K, E))
    landed = table.landed.numpy()
    assert landed[0] == 0 and landed[1] == 0 and landed[2] == 1 and landed[-1] == 1
    # the device form alone is close, not equal - the reason the table exists
    device = sim.camera_matrices(K, E)
    assert not torch.equal(device, cam) and torch.allclose(device, cam, rtol=1e-4, atol=1e-5)


def test_a_missed_calibration_is_served_by_the_device_form_then_exactly(sim):
    K, E = _skewed(9, 3)
    table = CalibrationTable(sim, 'cpu')
    table.prime(K[:4], E[:4])
    first = table.lookup(K, E)                                       # cameras 4..8 are new
    want = _oracle(K, E)
    assert np.array_equal(first[:4].numpy(), want[:4])
    assert torch.equal(first[4:], sim.camera_matrices(K[4:].contiguous(), E[4:].contiguous()))
    landed = table.landed.numpy()
    assert landed[0] == 5 and landed[1] == 5
    rows = landed[calibration.MISS_HEADER:calibration.MISS_HEADER + 5 * calibration.MISS_WORDS].reshape(5, -1)
    assert sorted(rows[:, calibration.KEY_WORDS].tolist()) == [4, 5, 6, 7, 8]
    second = table.lookup(K, E)                                      # first sighting of the five: candidates, not entries
    assert torch.equal(second, first) and table.stats['seen_once'] == 5 and table.stats['entries'] == 4
    third = table.lookup(K, E)                                       # they came back: filed from the list that landed
    assert np.array_equal(third.numpy(), want)
    assert table.stats['from_miss_lists'] == 5 and table.stats['entries'] == 9
    assert table.landed.numpy()[0] == 0


def test_calibrations_that_never_repeat_are_never_filed(sim, monkeypatch):
    """The reference's loader builds the extrinsics from each sample's own ego poses (fiery/data.py:172-209): every step
    brings new (K, [R | t]) pairs.  The table must not spend host LAPACK or uploads on them, must not fill up with one-off
    keys, and the output for a given input must not depend on the calls before it."""
    table = CalibrationTable(sim, 'cpu', slots=64)
    calls = []
    import fiery_amd.model as fm
    real = fm.host_camera_matrices
    monkeypatch.setattr(fm, 'host_camera_matrices', lambda *a: (calls.append(1), real(*a))[1])
    for step in range(12):
        K, E = _skewed(6, 100 + step)
        cam = table.lookup(K, E)
        assert torch.equal(cam, sim.camera_matrices(K, E))
    assert not calls and table.stats['entries'] == 0 and table.stats['from_miss_lists'] == 0
    assert len(table.candidates) <= table.max_candidates
    # a rig that DOES repeat is still filed, after the one-off keys, and served exactly
    K, E = _skewed(6, 7)
    for _ in range(3):
        cam = table.lookup(K, E)
    assert np.array_equal(cam.numpy(), _oracle(K, E)) and table.stats['entries'] == 6


def test_miss_list_overflow_and_duplicates(sim):
    """More new cameras than the list holds: the rest arrive with later calls; the same calibration at several cameras of one
    call makes one entry."""
    K, E = _skewed(6, 4)
    K, E = K.repeat(4, 1, 1), E.repeat(4, 1, 1)                      # 24 cameras, 6 distinct calibrations
    table = CalibrationTable(sim, 'cpu', miss_capacity=8)
    table.lookup(K, E)
    assert table.landed.numpy()[0] == 8 and table.landed.numpy()[1] == 24
    for _ in range(6):
        cam = table.lookup(K, E)
    assert np.array_equal(cam.numpy(), _oracle(K, E)) and table.stats['entries'] == 6


def test_full_table_and_singular_matrices_stay_on_the_device_form(sim):
    K, E = _skewed(40, 5)
    table = CalibrationTable(sim, 'cpu', slots=64)                   # holds 32 entries
    assert table.prime(K, E) == 32 and table.stats['rejected'] == 8
    cam = table.lookup(K, E)
    want = _oracle(K, E)
    held = np.array([calibration.key_words(K[i].numpy(), E[i].numpy())[0].tobytes() in table.known for i in range(40)])
    assert held.sum() == 32 and np.array_equal(cam.numpy()[held], want[held])
    assert torch.equal(cam[torch.from_numpy(~held)], sim.camera_matrices(K, E)[torch.from_numpy(~held)])
    singular = torch.zeros(1, 3, 3)
    fresh = CalibrationTable(sim, 'cpu')
    assert fresh.prime(torch.cat([K[:2], singular]), torch.cat([E[:2], E[:1]])) == 2 and fresh.stats['rejected'] == 1


def test_a_full_table_costs_the_host_nothing(sim, monkeypatch):
    """Calibrations that never repeat (a randomised K per sample) fill the table once; from then on every miss is served by
    the device form and the host neither inverts nor uploads anything."""
    from fiery_amd import model as model_module
    K, E = _skewed(40, 12)
    table = CalibrationTable(sim, 'cpu', slots=64)
    table.prime(K[:32], E[:32])
    assert len(table.known) == 32
    calls = []
    monkeypatch.setattr(model_module, 'host_camera_matrices', lambda *a: calls.append(1))
    for _ in range(3):
        cam = table.lookup(K, E)                                     # eight misses every time
    assert not calls and table.stats['from_miss_lists'] == 0
    assert torch.equal(cam[32:], sim.camera_matrices(K[32:].contiguous(), E[32:].contiguous()))
    assert table.prime(K[32:], E[32:]) == 0 and not calls


def test_pinhole_rigs_get_the_same_matrices_from_table_and_closed_form(sim):
    _, K, E, _ = make_inputs(1, 3, 6, with_image=False)
    K, E = K.reshape(-1, 3, 3), E.reshape(-1, 4, 4)
    table = CalibrationTable(sim, 'cpu')
    table.prime(K, E)
    assert torch.equal(table.lookup(K, E), sim.camera_matrices(K, E))


def test_model_table_mode_serves_get_geometry_exactly(sim):
    """`Fiery.get_geometry` in the 'table' mode: primed rigs exact; `prime_calibrations` is the set-up call.  The default is
    'device' (the same input gives the same output whatever was called before)."""
    from tests.helpers import tiny_cfg
    cfg = tiny_cfg('baseline.yml', bev=8)
    model = Fiery(cfg).eval()
    model._lib = sim
    assert model.camera_matrix_mode == 'device'
    model.camera_matrix_mode = 'table'
    K, E = _skewed(2, 6)
    K, E = K.view(1, 2, 3, 3), E.view(1, 2, 4, 4)
    assert model.prime_calibrations(K, E) == 2
    geo = model.get_geometry(K, E)
    want = ls.get_geometry(model.frustum.numpy(), K.numpy(), E.numpy())
    assert np.array_equal(geo.numpy(), want)
    model.camera_matrix_mode = 'device'
    assert not np.array_equal(model.get_geometry(K, E).numpy(), want)


# ----------------------------------------------------------------------------------------------------------------
# GPU tier
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_lookup_inside_a_captured_graph_is_bit_exact_for_skewed_intrinsics(hip):
    """The 'table' mode under a hipGraph: replays are bit-exact for every primed calibration written into the captured
    buffers; a calibration the table has never seen is served by the device form for the replays until its miss list has
    landed (here: one synchronise), then exactly - with no read-back on the replay path."""
    dev = torch.device('cuda:0')
    table = CalibrationTable(hip, dev)
    K1, E1 = _skewed(54, 7)
    K2, E2 = _skewed(54, 8)
    K3, E3 = _skewed(54, 9)
    table.prime(torch.cat([K1, K2]), torch.cat([E1, E2]))
    Kd, Ed = K1.to(dev), E1.to(dev)
    table.lookup(Kd, Ed)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cam = table.lookup(Kd, Ed)
    for K, E in ((K1, E1), (K2, E2)):
        Kd.copy_(K.to(dev)); Ed.copy_(E.to(dev))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(cam.cpu().numpy(), _oracle(K, E))
        assert table.landed[0].item() == 0
    Kd.copy_(K3.to(dev)); Ed.copy_(E3.to(dev))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cam, hip.camera_matrices(Kd, Ed))             # the device form, and a miss list on the host
    assert table.landed[0].item() == 54 and table.landed[1].item() == 54
    assert table.absorb_miss_lists() == 0 and table.stats['seen_once'] == 54    # first sighting: candidates only
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cam, hip.camera_matrices(Kd, Ed))
    assert table.absorb_miss_lists() == 54                                      # they came back: filed
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(cam.cpu().numpy(), _oracle(K3, E3))


@pytest.mark.gpu
def test_bev_forward_graph_table_mode_equals_host_mode_for_skewed_intrinsics(hip):
    """`bev_forward_graph` in the 'table' mode against the eager pass with `camera_matrix_mode = 'host'` (the reference's
    operators every call): same voxel ranks, same outputs - deterministic pooling, so bit for bit."""
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.synthetic import make_lifted_features
    from tests.helpers import randomise_weights
    dev = torch.device('cuda:0')
    cfg = get_preset_cfg('baseline.yml')
    torch.manual_seed(0)
    model = Fiery(cfg).eval()
    randomise_weights(model)
    model = model.to(dev)
    model.engine().pool_flags = native.POOL_DETERMINISTIC
    rf, n = model.receptive_field, 6
    _, K, E, ego = make_inputs(1, rf + model.n_future, n, with_image=False)
    gen = torch.Generator().manual_seed(11)
    K = K.clone()
    K[..., 0, 1] = 1.5 * torch.randn(K.shape[:3], generator=gen)
    K[..., 1, 0] = 0.2 * torch.randn(K.shape[:3], generator=gen)
    K[..., 2, 2] = 1.0 + 0.01 * torch.randn(K.shape[:3], generator=gen)
    fh, fw = cfg.IMAGE.FINAL_DIM[0] // 8, cfg.IMAGE.FINAL_DIM[1] // 8
    _, _, lifted = make_lifted_features(rf * n, 64, model.depth_channels, (fh, fw), seed=3)
    lifted = lifted.view(1, rf, n, 64, model.depth_channels, fh, fw).to(dev)
    noise = torch.zeros(1, 1, model.latent_dim, device=dev)
    Kd, Ed, egod = K.to(dev), E.to(dev), ego.to(dev)
    model.camera_matrix_mode = 'table'
    with torch.no_grad():
        got = {k: v.clone() for k, v in model.bev_forward_graph(lifted, Kd, Ed, egod, None, noise).items() if v is not None}
        got = {k: v.clone() for k, v in model.bev_forward_graph(lifted, Kd, Ed, egod, None, noise).items() if v is not None}
        geo_table = model.get_geometry(Kd[:, 0], Ed[:, 0])
        model.camera_matrix_mode = 'host'
        want = model.bev_forward(lifted, Kd, Ed, egod, None, noise)
        geo_host = model.get_geometry(Kd[:, 0], Ed[:, 0])
    assert torch.equal(geo_table, geo_host)
    assert np.array_equal(geo_host.cpu().numpy(), ls.get_geometry(model.frustum.cpu().numpy(), K[:, 0].numpy(), E[:, 0].numpy()))
    for k, v in got.items():
        assert torch.equal(v, want[k]), k
