"""Pose bookkeeping of the input pipeline: the camera extrinsics chain and the frame-to-frame ego-motion
(reference: fiery/data.py:165-207 `get_input_data`, :312-340 `get_future_egomotion`; fiery/utils/geometry.py:61-79, :82-106).

These are a handful of 4 x 4 products per sample, done once by the dataset worker on the host in the reference and here; they
produce the `extrinsics` (sensor -> flat lidar ego frame) and `future_egomotion` tensors the hot path consumes
(fiery/data.py:345-367).  The records are passed in as plain dictionaries with the nuScenes / Lyft field names ('rotation' =
quaternion (w, x, y, z), 'translation' = (x, y, z)), so the functions work on either SDK's tables without importing one.

The reference gets its quaternion algebra from `pyquaternion` (a dependency of nuscenes-devkit==1.1.0, environment.yml:20; absent
from this image): `Quaternion.rotation_matrix`, `.inverse`, `.yaw_pitch_roll` are restated below from that package's published
formulas, in float64 like the package; `tests/test_poses.py` checks them against scipy's independent `Rotation` and runs the
reference's own matrix helpers on top of them.
"""
import numpy as np
import torch


# ---- quaternion algebra (pyquaternion: quaternion.py `_normalise`, `rotation_matrix`, `inverse`, `yaw_pitch_roll`) -----------
def _unit(q):
    """`Quaternion._normalise`: scale to unit norm unless it already is one (|1 - |q|^2| < 1e-14) or is zero."""
    q = np.asarray(q, dtype=np.float64)
    if q.shape != (4,):
        raise ValueError('a quaternion is four numbers (w, x, y, z)')
    s = float(np.dot(q, q))
    if abs(1.0 - s) < 1e-14:
        return q
    n = np.sqrt(s)
    return q / n if n > 0 else q


def quaternion_rotation_matrix(q):
    """3 x 3 rotation of the unit quaternion (w, x, y, z): the lower-right block of Q(q) . conj(Qbar(q))^T."""
    w, x, y, z = _unit(q)
    q_matrix = np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])
    q_bar_matrix = np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])
    return np.dot(q_matrix, q_bar_matrix.conj().transpose())[1:][:, 1:]


def quaternion_inverse(q):
    """conj(q) / |q|^2 (zero quaternions have no inverse)."""
    q = np.asarray(q, dtype=np.float64)
    s = float(np.dot(q, q))
    if s <= 0:
        raise ZeroDivisionError('a zero quaternion cannot be inverted')
    return np.array([q[0], -q[1], -q[2], -q[3]]) / s


def quaternion_yaw(q):
    """First of `yaw_pitch_roll`: atan2(2 (w z - x y), 1 - 2 (y^2 + z^2)) of the normalised quaternion."""
    w, x, y, z = _unit(q)
    return np.arctan2(2 * (w * z - x * y), 1 - 2 * (y ** 2 + z ** 2))


def _rigid(rotation, translation_column):
    return np.vstack([np.hstack((rotation, translation_column)), np.array([0, 0, 0, 1])])


# ---- fiery/data.py:165-207 ---------------------------------------------------------------------------------------------------
def lidar_to_world(lidar_pose):
    """The 'flat' lidar ego pose: only the yaw of the ego rotation is kept (fiery/data.py:172-181)."""
    yaw = quaternion_yaw(lidar_pose['rotation'])
    flat = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    return _rigid(quaternion_rotation_matrix(flat), np.array(lidar_pose['translation'])[:, None])


def sensor_to_lidar(lidar_pose, camera_pose, calibrated_sensor):
    """(4, 4) float32 extrinsics of one camera: sensor -> flat lidar ego frame (fiery/data.py:183-207).

    lidar_pose / camera_pose: the 'ego_pose' records of the LIDAR_TOP and of the camera sample_data; calibrated_sensor: the camera's
    'calibrated_sensor' record."""
    to_world = lidar_to_world(lidar_pose)
    rot = quaternion_rotation_matrix(quaternion_inverse(camera_pose['rotation']))
    world_to_car = _rigid(rot, rot @ (-np.array(camera_pose['translation'])[:, None]))
    car_to_sensor = np.linalg.inv(_rigid(quaternion_rotation_matrix(calibrated_sensor['rotation']),
                                         np.array(calibrated_sensor['translation'])[:, None]))
    lidar_to_sensor = car_to_sensor @ world_to_car @ to_world
    return torch.from_numpy(np.linalg.inv(lidar_to_sensor)).float()


def camera_rig_extrinsics(lidar_pose, camera_poses, calibrated_sensors):
    """(N, 4, 4) extrinsics and (N, 3, 3) intrinsics of one time step (fiery/data.py:183-228 without the image work)."""
    extrinsics = torch.stack([sensor_to_lidar(lidar_pose, p, s) for p, s in zip(camera_poses, calibrated_sensors)])
    intrinsics = torch.stack([torch.Tensor(s['camera_intrinsic']) for s in calibrated_sensors])
    return intrinsics, extrinsics


# ---- fiery/utils/geometry.py:61-79, fiery/data.py:312-340 ----------------------------------------------------------------------
def convert_egopose_to_matrix_numpy(egopose):
    """float32 (4, 4) of an 'ego_pose' record (geometry.py:61-68)."""
    m = np.zeros((4, 4), dtype=np.float32)
    m[:3, :3] = quaternion_rotation_matrix(egopose['rotation'])
    m[:3, 3] = np.array(egopose['translation'])
    m[3, 3] = 1.0
    return m


def invert_matrix_egopose_numpy(egopose):
    """Inverse of a rigid float32 (4, 4) (geometry.py:71-79)."""
    inv = np.zeros((4, 4), dtype=np.float32)
    rotation, translation = egopose[:3, :3], egopose[:3, 3]
    inv[:3, :3] = rotation.T
    inv[:3, 3] = -np.dot(rotation.T, translation)
    inv[3, 3] = 1.0
    return inv


def mat2pose_vec(matrix):
    """(..., 4, 4) -> (..., 6) translation + Euler angles (geometry.py:82-106)."""
    rotx = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    cosy = torch.sqrt(matrix[..., 1, 2] ** 2 + matrix[..., 2, 2] ** 2)
    roty = torch.atan2(matrix[..., 0, 2], cosy)
    rotz = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rotx, roty, rotz), dim=-1)), dim=-1)


def future_egomotion(egopose_t0, egopose_t1=None):
    """(1, 6) motion from frame t0 to t1 as the hot path expects it (fiery/data.py:312-340); identity when there is no next
    frame in the same scene (`egopose_t1` None)."""
    motion = np.eye(4, dtype=np.float32)
    if egopose_t1 is not None:
        t0 = convert_egopose_to_matrix_numpy(egopose_t0)
        t1 = convert_egopose_to_matrix_numpy(egopose_t1)
        motion = invert_matrix_egopose_numpy(t1).dot(t0)
        motion[3, :3] = 0.0
        motion[3, 3] = 1.0
    return mat2pose_vec(torch.Tensor(motion).float()).unsqueeze(0)
