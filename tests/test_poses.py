"""Pose bookkeeping of the input pipeline (`fiery_amd/poses.py`): the quaternion algebra restated from pyquaternion against
scipy's independent `Rotation`, and the extrinsics chain / future ego-motion against the reference's own dataset methods run
here on made-up nuScenes records (fiery/data.py:150-228, :312-340)."""
import numpy as np
import pytest
import torch

from fiery_amd import poses


def _quaternions(n, seed=0, unit=True):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4))
    if unit:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def test_rotation_matrix_inverse_and_yaw_against_scipy():
    from scipy.spatial.transform import Rotation
    for q in list(_quaternions(50)) + list(_quaternions(10, seed=1, unit=False) * 3.0):
        w, x, y, z = q
        rot = Rotation.from_quat([x, y, z, w])
        assert np.abs(poses.quaternion_rotation_matrix(q) - rot.as_matrix()).max() < 1e-14
        inv = poses.quaternion_inverse(q)
        assert np.abs(poses.quaternion_rotation_matrix(inv) - rot.inv().as_matrix()).max() < 1e-14
        # the package's yaw is the z angle of R = R_x(roll) R_y(pitch) R_z(yaw)
        assert abs(poses.quaternion_yaw(q) - rot.as_euler('zyx')[0]) < 1e-12
    with pytest.raises(ZeroDivisionError):
        poses.quaternion_inverse([0, 0, 0, 0])
    with pytest.raises(ValueError):
        poses.quaternion_rotation_matrix([1, 0, 0])


def test_flat_lidar_pose_keeps_only_the_yaw():
    q = _quaternions(1, seed=3)[0]
    m = poses.lidar_to_world(dict(rotation=list(q), translation=[10.0, -4.0, 1.5]))
    yaw = poses.quaternion_yaw(q)
    want = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    assert np.abs(m[:3, :3] - want).max() < 1e-15
    assert np.array_equal(m[:3, 3], [10.0, -4.0, 1.5]) and np.array_equal(m[3], [0, 0, 0, 1])


def test_future_egomotion_identity_without_a_next_frame():
    v = poses.future_egomotion(dict(rotation=[1, 0, 0, 0], translation=[1, 2, 3]))
    assert v.shape == (1, 6) and torch.equal(v, torch.zeros(1, 6))


# ---- against the reference's dataset code ----------------------------------------------------------------------------
class _Tables:
    """`nusc.get(table, token)` over dictionaries."""

    def __init__(self, tables):
        self.tables = tables

    def get(self, table, token):
        return self.tables[table][token]


def _scene(n_cams, seed, tmp_path, image_hw):
    from PIL import Image
    rng = np.random.default_rng(seed)
    cams = [f'CAM_{i}' for i in range(n_cams)]
    tables = dict(sample_data={}, ego_pose={}, calibrated_sensor={})
    recs = []
    for t in range(2):
        base = rng.standard_normal(4)
        base[0] += 3.0                                   # mostly yaw-free ego rotations would hide errors: keep them generic
        lidar_pose = dict(rotation=list(base / np.linalg.norm(base) * (1.0 + 1e-9 * t)),      # not exactly unit, like real tables
                          translation=list(rng.uniform(-500, 500, 3)))
        tables['ego_pose'][f'lp{t}'] = lidar_pose
        tables['sample_data'][f'lidar{t}'] = dict(ego_pose_token=f'lp{t}')
        data = {'LIDAR_TOP': f'lidar{t}'}
        for c in cams:
            q = base + 0.01 * rng.standard_normal(4)
            tables['ego_pose'][f'cp{t}{c}'] = dict(rotation=list(q / np.linalg.norm(q)),
                                                   translation=list(np.array(lidar_pose['translation']) + rng.uniform(-0.5, 0.5, 3)))
            qs = rng.standard_normal(4)
            tables['calibrated_sensor'][f'cs{t}{c}'] = dict(
                rotation=list(qs / np.linalg.norm(qs)), translation=list(rng.uniform(-2, 2, 3)),
                camera_intrinsic=[[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
            name = f'img{t}{c}.png'
            Image.fromarray(rng.integers(0, 256, (*image_hw, 3), dtype=np.uint8)).save(tmp_path / name)
            tables['sample_data'][f'sd{t}{c}'] = dict(ego_pose_token=f'cp{t}{c}', calibrated_sensor_token=f'cs{t}{c}', filename=name)
            data[c] = f'sd{t}{c}'
        recs.append(dict(data=data, scene_token='scene'))
    return cams, tables, recs


@pytest.mark.needs_reference
def test_extrinsics_chain_and_egomotion_against_the_reference_dataset_methods(tmp_path):
    from types import SimpleNamespace
    from oracle.ref_shims import load_reference_data
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.images import get_resizing_and_cropping_parameters, update_intrinsics
    ref = load_reference_data()
    cfg = get_preset_cfg('baseline.yml', ['IMAGE.ORIGINAL_HEIGHT', '90', 'IMAGE.ORIGINAL_WIDTH', '160', 'IMAGE.FINAL_DIM', '(22, 48)',
                                          'IMAGE.TOP_CROP', '4', 'IMAGE.NAMES', "['CAM_0', 'CAM_1', 'CAM_2']"])
    cams, tables, recs = _scene(3, 7, tmp_path, (90, 160))
    fake = SimpleNamespace(cfg=cfg, nusc=_Tables(tables), dataroot=str(tmp_path), ixes=recs)
    cls = ref.data.FuturePredictionDataset
    fake.augmentation_parameters = cls.get_resizing_and_cropping_parameters(fake)
    fake.normalise_image = ref.data.torchvision.transforms.Compose(
        [ref.data.torchvision.transforms.ToTensor(),
         ref.data.torchvision.transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    aug = get_resizing_and_cropping_parameters(cfg)
    assert {k: tuple(v) if isinstance(v, (list, tuple)) else v for k, v in aug.items()} == \
        {k: tuple(v) if isinstance(v, (list, tuple)) else v for k, v in fake.augmentation_parameters.items()}
    for t, rec in enumerate(recs):
        _, want_k, want_e = cls.get_input_data(fake, rec)                       # (1, N, 3, 3), (1, N, 4, 4)
        lidar_pose = tables['ego_pose'][f'lp{t}']
        cam_poses = [tables['ego_pose'][f'cp{t}{c}'] for c in cams]
        sensors = [tables['calibrated_sensor'][f'cs{t}{c}'] for c in cams]
        k, e = poses.camera_rig_extrinsics(lidar_pose, cam_poses, sensors)
        k = torch.stack([update_intrinsics(ki, aug['crop'][1], aug['crop'][0], scale_width=aug['scale_width'],
                                           scale_height=aug['scale_height']) for ki in k])
        assert e.dtype == torch.float32 and e.shape == (3, 4, 4)
        # float64 chains rounded to float32 at the end: the two quaternion implementations differ by a few float64 ulps
        assert (e - want_e[0]).abs().max().item() <= 1e-6 * max(1.0, want_e.abs().max().item())
        assert torch.equal(k, want_k[0])
    want = cls.get_future_egomotion(fake, recs[0], 0)
    got = poses.future_egomotion(tables['ego_pose']['lp0'], tables['ego_pose']['lp1'])
    assert got.shape == want.shape == (1, 6)
    # translations of hundreds of metres in float32 (geometry.py:62): one float32 ulp of the inputs is 3e-5 m
    assert (got - want).abs().max().item() <= 1e-4
    last = cls.get_future_egomotion(fake, recs[1], 1)                          # no next frame: identity
    assert torch.equal(last, poses.future_egomotion(tables['ego_pose']['lp1'], None))
    # the reference's own matrix helpers on the reference's numbers, and ours on the same
    m = ref.geometry.convert_egopose_to_matrix_numpy(tables['ego_pose']['lp0'])
    assert np.array_equal(ref.geometry.invert_matrix_egopose_numpy(m), poses.invert_matrix_egopose_numpy(m))
    assert np.abs(m - poses.convert_egopose_to_matrix_numpy(tables['ego_pose']['lp0'])).max() <= 1e-6 * np.abs(m).max()
