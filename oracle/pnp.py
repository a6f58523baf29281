"""ORACLE (test infrastructure): float64 numpy restatement of the cuboid PnP stage.

PARITY UNPINNED.  The arithmetic of this stage is not in the reference tree: the reference calls
``cv2.solvePnPGeneric(..., flags=SOLVEPNP_ITERATIVE)`` and ``cv2.projectPoints`` from the un-vendored,
un-pinned dependency ``opencv-python>=4.5.3.56`` (/root/reference/requirements.txt:11; call sites
/root/reference/src/lib/utils/pnp/cuboid_pnp_solver.py:159-171, :203-205) and no reference test pins
its results; OpenCV is not installed here.  This file restates OpenCV calib3d's published algorithm
for SOLVEPNP_ITERATIVE on >= 6 non-planar points (cvFindExtrinsicCameraParams2: normalise by K,
DLT on the 2N x 12 system via the smallest eigenvector of L^T L, det sign fix, SVD orthogonalisation
R = U V^T with t rescaled by |R|/|R_raw|, Rodrigues, then CvLevMarq: <= 20 iterations, eps =
FLT_EPSILON, lambda = 10^k with k starting at -3, JtJ diagonal scaled by (1 + lambda)), and anchors
correctness on construction: noise-free known poses are recovered to 1e-8, noisy ones agree with an
independent scipy least-squares minimiser (tests/test_pnp_oracle.py).

Everything around the solver follows the reference's own Python:
  cuboid vertices        cuboid_objectron.py:80-110
  point filtering/order  cuboid_pnp_solver.py:141-157
  z < 0 rejection        cuboid_pnp_solver.py:207-220
  OpenGL conversion      cuboid_pnp_solver.py:179-196
  quaternion             cuboid_pnp_solver.py:241-247 (pyrr axis-angle, xyzw)
  pnp_shell packaging    cuboid_pnp_shell.py:11-93
"""
import numpy as np

FLT_EPSILON = 1.1920928955078125e-07
DBL_EPSILON = 2.220446049250313e-16


def cuboid_vertices(size3d):
    """cuboid_objectron.py:80-110: x = width, y = height, z = depth, centred at the origin."""
    w, h, d = [float(v) for v in size3d]
    r, l = w / 2.0, -w / 2.0
    t, b = h / 2.0, -h / 2.0
    f, re = d / 2.0, -d / 2.0
    return np.array([[l, b, re], [l, b, f], [l, t, re], [l, t, f],
                     [r, b, re], [r, b, f], [r, t, re], [r, t, f]], dtype=np.float64)


def rodrigues_to_matrix(r, jac=False):
    """cv::Rodrigues vector -> matrix (+ 3x9 jacobian dR/dr, R row-major)."""
    r = np.asarray(r, np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < DBL_EPSILON:
        R = np.eye(3)
        J = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [0, 0, 1, 0, 0, 0, -1, 0, 0], [0, -1, 0, 1, 0, 0, 0, 0, 0]],
                     np.float64)
        return (R, J) if jac else R
    c, s = np.cos(theta), np.sin(theta)
    c1 = 1.0 - c
    it = 1.0 / theta
    a = r * it
    rrt = np.outer(a, a)
    rx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = c * np.eye(3) + c1 * rrt + s * rx
    if not jac:
        return R
    I = np.eye(3).reshape(9)
    x, y, z = a
    drrt = np.array([[x + x, y, z, y, 0, 0, z, 0, 0], [0, x, 0, x, y + y, z, 0, z, 0], [0, 0, x, 0, 0, y, x, y, z + z]])
    drx = np.array([[0, 0, 0, 0, 0, -1, 0, 1, 0], [0, 0, 1, 0, 0, 0, -1, 0, 0], [0, -1, 0, 1, 0, 0, 0, 0, 0]], np.float64)
    J = np.zeros((3, 9))
    for i in range(3):
        ri = a[i]
        a0, a1, a2, a3, a4 = -s * ri, (s - 2 * c1 * it) * ri, c1 * it, (c - s * it) * ri, s * it
        J[i] = a0 * I + a1 * rrt.reshape(9) + a2 * drrt[i] + a3 * rx.reshape(9) + a4 * drx[i]
    return R, J


def matrix_to_rodrigues(R):
    """cv::Rodrigues matrix -> vector (re-orthogonalises by SVD first)."""
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return np.array([rx, ry, rz]) * (theta / (2 * s))


def project_points(obj, rvec, tvec, Kmat, jac=False):
    """cv::projectPoints without distortion (+ d(u,v)/d(rvec,tvec) as a 2N x 6 matrix)."""
    fx, fy, cx, cy = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2]
    if jac:
        R, dRdr = rodrigues_to_matrix(rvec, True)
    else:
        R = rodrigues_to_matrix(rvec)
    P = obj @ R.T + np.asarray(tvec, np.float64).reshape(1, 3)
    z = np.where(P[:, 2] != 0, 1.0 / np.where(P[:, 2] != 0, P[:, 2], 1.0), 1.0)
    x, y = P[:, 0] * z, P[:, 1] * z
    uv = np.stack([fx * x + cx, fy * y + cy], axis=1)
    if not jac:
        return uv
    N = obj.shape[0]
    J = np.zeros((2 * N, 6))
    for j in range(3):
        dx0 = obj @ dRdr[j, 0:3]
        dy0 = obj @ dRdr[j, 3:6]
        dz0 = obj @ dRdr[j, 6:9]
        J[0::2, j] = fx * z * (dx0 - x * dz0)
        J[1::2, j] = fy * z * (dy0 - y * dz0)
    J[0::2, 3] = fx * z
    J[0::2, 5] = -fx * x * z
    J[1::2, 4] = fy * z
    J[1::2, 5] = -fy * y * z
    return uv, J


def dlt_init(obj, mn):
    """Non-planar initialisation of cvFindExtrinsicCameraParams2 (needs >= 6 points)."""
    N = obj.shape[0]
    L = np.zeros((2 * N, 12))
    for i in range(N):
        X, Y, Z = obj[i]
        x, y = -mn[i, 0], -mn[i, 1]
        L[2 * i] = [X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x]
        L[2 * i + 1] = [0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y]
    LL = L.T @ L
    w, V = np.linalg.eigh(LL)  # ascending eigenvalues
    RRt = V[:, 0].reshape(3, 4)
    RR = RRt[:, :3]
    tt = RRt[:, 3]
    if np.linalg.det(RR) < 0:
        RR, tt = -RR, -tt
    sc = np.linalg.norm(RR)
    U, _, Vt = np.linalg.svd(RR)
    R = U @ Vt
    t = tt * (np.linalg.norm(R) / sc)
    return matrix_to_rodrigues(R), t


def is_planar(obj):
    Mc = obj.mean(axis=0)
    MM = (obj - Mc).T @ (obj - Mc)
    w = np.linalg.svd(MM, compute_uv=False)
    return w[2] / w[1] < 1e-3


def solve_pnp_iterative(obj, img, Kmat, max_iter=20, eps=FLT_EPSILON, return_iters=False):
    """SOLVEPNP_ITERATIVE for >= 6 non-planar points, zero distortion.  Returns (ok, rvec, tvec)."""
    obj = np.asarray(obj, np.float64)
    img = np.asarray(img, np.float64)
    Kmat = np.asarray(Kmat, np.float64)
    N = obj.shape[0]
    if N < 6 or is_planar(obj):
        raise NotImplementedError("oracle covers the non-planar DLT branch (>= 6 points)")
    fx, fy, cx, cy = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2]
    mn = np.stack([(img[:, 0] - cx) / fx, (img[:, 1] - cy) / fy], axis=1)
    r, t = dlt_init(obj, mn)
    param = np.concatenate([r, t])
    # ---- CvLevMarq state machine (completeSymmFlag = true, DECOMP_SVD) ----
    lambda_lg10 = -3
    iters = 0
    prev_param = param.copy()
    prev_err_norm = None
    state = "CALC_J"
    JtJ = JtErr = None
    while True:
        if state == "CALC_J":
            uv, J = project_points(obj, param[:3], param[3:], Kmat, True)
            err = (uv - img).reshape(-1)
            JtJ = J.T @ J
            JtErr = J.T @ err
            prev_param = param.copy()
            if iters == 0:
                prev_err_norm = np.linalg.norm(err)
            param = _lm_step(JtJ, JtErr, prev_param, lambda_lg10)
            state = "CHECK_ERR"
            continue
        # CHECK_ERR
        uv = project_points(obj, param[:3], param[3:], Kmat)
        err_norm = np.linalg.norm((uv - img).reshape(-1))
        if err_norm > prev_err_norm:
            lambda_lg10 += 1
            if lambda_lg10 <= 16:
                param = _lm_step(JtJ, JtErr, prev_param, lambda_lg10)
                continue
        lambda_lg10 = max(lambda_lg10 - 1, -16)
        iters += 1
        denom = np.linalg.norm(prev_param)
        rel = np.linalg.norm(param - prev_param) / (denom if denom > 0 else 1.0)
        if iters >= max_iter or rel < eps:
            break
        prev_err_norm = err_norm
        state = "CALC_J"
    if return_iters:
        return True, param[:3].copy(), param[3:].copy(), iters
    return True, param[:3].copy(), param[3:].copy()


def _lm_step(JtJ, JtErr, prev_param, lambda_lg10):
    lam = np.exp(lambda_lg10 * np.log(10.0))
    A = JtJ.copy()
    A[np.diag_indices_from(A)] *= 1.0 + lam
    delta = np.linalg.lstsq(A, JtErr, rcond=None)[0]
    return prev_param - delta


def axis_angle_quat_xyzw(rvec):
    """cuboid_pnp_solver.py:241-247 + pyrr.Quaternion.from_axis_rotation (normalised axis)."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    theta = np.sqrt(rvec[0] * rvec[0] + rvec[1] * rvec[1] + rvec[2] * rvec[2])
    axis = rvec / theta
    axis = axis / np.linalg.norm(axis)
    h = theta * 0.5
    return np.array([np.sin(h) * axis[0], np.sin(h) * axis[1], np.sin(h) * axis[2], np.cos(h)])


def quat_xyzw_to_matrix(q):
    """scipy Rotation.from_quat(q).as_matrix() (normalises q)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def solve_cuboid_pnp(points2d, scale, Kmat, opencv_return=False):
    """CuboidPNPSolver.solve_pnp (cuboid_pnp_solver.py:91-239) for one detection.
    points2d: (8*n, 2), entries with x or y < -5000 are invalid.  scale: relative cuboid size.
    Returns dict(location, quaternion_xyzw, projected_points (8x2), rvec, tvec, reproj_err) or None."""
    pts = np.asarray(points2d, np.float64).reshape(-1, 2)
    verts = cuboid_vertices(np.asarray(scale, np.float64) / scale[1])  # cuboid_pnp_shell.py:12
    n_per = len(pts) / 8
    o2, o3 = [], []
    for i in range(len(pts)):
        if pts[i, 0] < -5000 or pts[i, 1] < -5000:
            continue
        o2.append(pts[i])
        o3.append(verts[int(i // n_per)])
    o2 = np.array(o2, dtype=float)
    o3 = np.array(o3, dtype=float)
    if len(o2) < 4:
        return None
    if len(o2) < 6:
        raise NotImplementedError("EPnP branch (4-5 valid points) is not restated")
    ok, rvec, tvec = solve_pnp_iterative(o3, o2, Kmat)
    proj_v = project_points(o3, rvec, tvec, Kmat)
    reproj = float(np.linalg.norm(proj_v - o2) / np.sqrt(2 * len(o2)))
    R = rodrigues_to_matrix(rvec)
    M3 = np.array([[0, 1, 0], [1, 0, 0], [0, 0, -1]], np.float64)
    R_gl = M3 @ R
    t_gl = M3 @ tvec
    rvec_gl = matrix_to_rodrigues(R_gl)
    projected = project_points(verts, rvec, tvec, Kmat)
    if tvec[2] < 0:
        return None
    if opencv_return:
        loc, quat = tvec.copy(), axis_angle_quat_xyzw(rvec)
    else:
        loc, quat = t_gl, axis_angle_quat_xyzw(rvec_gl)
    return {"location": loc, "quaternion_xyzw": quat, "projected_points": projected, "rvec": rvec, "tvec": tvec,
            "reproj_err": reproj}


def pnp_shell(points2d, scale, Kmat, width, height, category="cup", kps=None, opencv_return=False):
    """cuboid_pnp_shell.py:11-93.  Returns None when the detection is dropped, else a dict with
    kps_pnp (9x2 normalised), kps_3d_cam (9x3), location, quaternion_xyzw, projected_cuboid."""
    sol = solve_cuboid_pnp(points2d, scale, Kmat, opencv_return)
    if sol is None:
        return None
    verts = cuboid_vertices(np.asarray(scale, np.float64) / scale[1])
    ori = quat_xyzw_to_matrix(sol["quaternion_xyzw"])
    cam = verts @ ori.T + np.asarray(sol["location"]).reshape(1, 3)
    cam = np.vstack([cam.mean(axis=0, keepdims=True), cam])
    proj = sol["projected_points"]
    proj = np.vstack([proj.mean(axis=0, keepdims=True), proj]).copy()
    proj[:, 0] /= width
    proj[:, 1] /= height
    if category not in ("bike", "laptop", "shoe"):
        thr = 6 if category in ("book", "chair", "cereal_box") else 3
        nv = sum(1 for p in proj if p[0] < 0 or p[0] > 1 or p[1] < 0 or p[1] > 1)
        if nv >= thr:
            return None
    if not (0 < proj[0][0] < 1 and 0 < proj[0][1] < 1):
        return None
    out = dict(sol)
    out["kps_pnp"] = proj
    out["kps_3d_cam"] = cam
    return out
