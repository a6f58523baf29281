"""``CuboidPNPSolver`` with the reference's interface (utils/pnp/cuboid_pnp_solver.py:13-247).  The solve
itself (``cv2.solvePnPGeneric`` + ``cv2.projectPoints`` in the reference) runs in the batched HIP kernel
``cp_pnp_solve``; ``solve_pnp_batch`` is the added batched entry point."""
import numpy as np
import torch

from centerpose_amd import hip as _hip


def _cam4(K):
    K = np.asarray(K, np.float64).reshape(3, 3)
    return [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]


def solve_pnp_batch(points_list, scales, camera_matrices, device='cuda'):
    """points_list: N arrays (8 or 16, 2); scales: N x 3; camera_matrices: one 3x3 or N of them.
    Returns the raw [N, 40] float64 result rows (layout in include/centerpose_hip.h)."""
    n = len(points_list)
    if n == 0:
        return np.zeros((0, _hip.PNP_STRIDE))
    pts = np.stack([np.asarray(p, np.float32).reshape(-1, 2) for p in points_list])
    cams = np.asarray(camera_matrices, np.float64)
    cam = np.tile(_cam4(cams), (n, 1)) if cams.ndim == 2 else np.array([_cam4(k) for k in cams])
    out = _hip.pnp_solve(torch.from_numpy(pts).to(device), torch.from_numpy(np.asarray(scales, np.float32)).to(device),
                         torch.from_numpy(cam).to(device))
    return out.cpu().numpy()


class CuboidPNPSolver(object):
    def __init__(self, object_name="", scaling_factor=1, camera_intrinsic_matrix=None, cuboid3d=None,
                 dist_coeffs=np.zeros((4, 1)), min_required_points=4):
        self.object_name = object_name
        self.min_required_points = max(4, min_required_points)
        self.scaling_factor = scaling_factor
        self._camera_intrinsic_matrix = camera_intrinsic_matrix if camera_intrinsic_matrix is not None \
            else np.zeros((3, 3))
        self._cuboid3d = cuboid3d
        self._dist_coeffs = dist_coeffs

    def set_camera_intrinsic_matrix(self, new_intrinsic_matrix):
        self._camera_intrinsic_matrix = new_intrinsic_matrix

    def set_dist_coeffs(self, dist_coeffs):
        self._dist_coeffs = dist_coeffs

    def solve_pnp(self, cuboid2d_points, pnp_algorithm=None, OPENCV_RETURN=False, fail_if_projected_diff_exceeds=250,
                  fail_if_projected_value_exceeds=1e5, verbose=False):
        """Returns (location, quaternion_xyzw, projected_points, reprojectionError) like the reference
        (:91-239); location / quaternion are None when the detection must be dropped."""
        if np.any(np.asarray(self._dist_coeffs) != 0):
            raise NotImplementedError("lens distortion is not modelled (the reference always passes zeros)")
        size = np.asarray(self._cuboid3d.size3d, np.float64)
        r = solve_pnp_batch([cuboid2d_points], [size], self._camera_intrinsic_matrix)[0]
        status = int(r[0])
        projected = cuboid2d_points
        err = None
        if status >= 1:
            projected = r[8:24].reshape(8, 2).copy()
            err = float(r[7])
        if status != 1:
            return None, None, projected, err
        if OPENCV_RETURN:
            return list(r[4:7]), r[24:28].copy(), projected, err
        return list(r[28:31]), r[31:35].copy(), projected, err
