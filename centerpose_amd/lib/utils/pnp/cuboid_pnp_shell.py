"""``pnp_shell`` with the reference's signature and return tuple (utils/pnp/cuboid_pnp_shell.py:11-93)."""
import numpy as np

from .cuboid_objectron import Cuboid3d
from .cuboid_pnp_solver import CuboidPNPSolver


def _quat_to_matrix(q):
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)  # scipy Rotation.from_quat normalises
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def finish_detection(opt, meta, bbox, scale, location, quaternion, projected_points):
    """Everything pnp_shell does after the solve (:26-91); shared with the batched detector path."""
    if location is None:
        return None
    cuboid3d = Cuboid3d(1 * np.array(scale) / scale[1])
    bbox['location'] = location
    bbox['quaternion_xyzw'] = quaternion
    bbox['projected_cuboid'] = projected_points
    pose = np.identity(4)
    pose[:3, :3] = _quat_to_matrix(quaternion)
    pose[:3, 3] = location
    obj = np.array(cuboid3d.get_vertices())
    cam = (pose @ np.hstack((obj, np.ones((obj.shape[0], 1)))).T)[:3, :].T
    cam = np.insert(cam, 0, np.mean(cam, axis=0), axis=0)
    bbox['kps_3d_cam'] = cam
    projected_points = np.insert(projected_points, 0, np.mean(projected_points, axis=0), axis=0)
    projected_points[:, 0] = projected_points[:, 0] / meta['width']
    projected_points[:, 1] = projected_points[:, 1] / meta['height']
    bbox['kps_pnp'] = projected_points
    if opt.c not in ['bike', 'laptop', 'shoe']:
        if opt.c in ['book', 'chair', 'cereal_box']:
            thresh = 6
        if opt.c in ['camera', 'bottle', 'cup']:
            thresh = 3
        n_out = sum(1 for p in projected_points if p[0] < 0 or p[0] > 1 or p[1] < 0 or p[1] > 1)
        if n_out >= thresh:
            return None
    p0 = projected_points[0]
    if not (p0[0] > 0 and p0[0] < 1 and p0[1] > 0 and p0[1] < 1):
        return None
    points = [(x[0], x[1]) for x in np.array(bbox['kps']).reshape(-1, 2)]
    points_ori = np.insert(points, 0, np.mean(points, axis=0), axis=0)
    points_ori[:, 0] = points_ori[:, 0] / meta['width']
    points_ori[:, 1] = points_ori[:, 1] / meta['height']
    return projected_points, cam, np.array(bbox['obj_scale']), points_ori, bbox


def pnp_shell(opt, meta, bbox, points_filtered, scale, OPENCV_RETURN=False):
    cuboid3d = Cuboid3d(1 * np.array(scale) / scale[1])
    solver = CuboidPNPSolver(opt.c, cuboid3d=cuboid3d)
    solver.set_camera_intrinsic_matrix(meta['camera_matrix'])
    location, quaternion, projected_points, _ = solver.solve_pnp(points_filtered, OPENCV_RETURN=OPENCV_RETURN)
    return finish_detection(opt, meta, bbox, scale, location, quaternion, projected_points)
